#!/usr/bin/env python3
"""bench.py — headline benchmark of the fused perception->reasoning hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input: G x B concurrent cortex cycles, each = (640x640 frame
detect + 1 s PCM VAD/ASR, when those streams are enabled) + 64-token prompt prefill + 128-token greedy Mistral-7B Q4_K_M decode.
G x B is stated in config (the weight stream is shared by the cycles decoded together).

    python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches one rank per GPU with torch.distributed.run; by default ranks are independent replicas of the cycle
batch (no data-path collective, weak scaling); --placement model-per-gpu gives the LLM and the perception streams their own GPUs
(SURVEY.md §8e).  Prints ONE JSON line on rank 0.

What the line carries besides the contract keys (all measured live in this run, rank 0, N = 1):
  roofline            the W4A8 launch set of one decode step at the headline's rows per pass: WEIGHT bytes / sum of launch times / 8 TB/s
                      (SURVEY.md §8d), HIP events on the session stream; `traffic` = PMC HBM bytes per launch from profiles/ when that
                      file was produced from THIS kernel source (sha stamp), else null
  roofline_attention  the k_attention launch set, bounded by KV-cache bytes
  north_star_point    a second, short timed run at 3 x 16 rows per pass: the configuration that meets north_star's
                      ">= 30 cycles/s at >= 40 % of the HBM roofline" together, reported beside the throughput-optimal headline
  reference_abi_b1    batch 1 through the reference's own entry points only (tk_llm_runner_*, tk_cortex_*)
  prompt_processing   one prompt of 64 / 448 / 2 048 tokens through the session's prefill (time to the first token's logits)
  long_context_decode one sequence decoding at ~2 100 and ~4 000 cached positions (ms per token)
  reference_abi_batched_cortex   K cortex handles on one model file, one DATA-DEPENDENT cycle each through tk_cortex_* only
  cpu_baseline        one whole fused cycle on the CPU oracle ("port"), at the box's core share and at 1 core
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
INT8_PEAK_TOPS = 5000.0  # dense int8 MFMA, 2x the bf16 rate (same guide, matrix cores)
Q4K_BPW, Q6K_BPW = 144.0 / 256.0, 210.0 / 256.0
METRIC = "cortex cycles/sec (frame+1s audio+128 tok)"
DTYPE = "int8 x q4_K/q6_K (i32 acc, f32 scales)"


def splitmix_tokens(seed, n, lo, hi):
    rng = np.random.default_rng(seed)
    return rng.integers(lo, hi, n).astype(np.int32)


def kernel_source_sha():
    """identifies the W4A8 kernel source a PMC summary under profiles/ was collected from"""
    h = hashlib.sha256()
    for rel in ("trackiellm_amd/csrc/llm/tk_llm_kernels.hip", "trackiellm_amd/csrc/llm/tk_llm_layout.h"):
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()[:16]


def newest_pmc(name):
    """the newest profiles/rNN_pmc_<name>.json (the round prefix sorts), or a path that does not exist"""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_%s.json" % name)))
    return found[-1] if found else os.path.join(ROOT, "profiles", "none_pmc_%s.json" % name)


def q6_layers(n_layer):
    return [l for l in range(n_layer) if l < n_layer // 8 or l >= 7 * n_layer // 8 or (l - n_layer // 8) % 3 == 2]


class CycleBench:
    """G decode groups x B cycles; one step() = every stream once over G x B cycles: the detector / ASR / VAD streams and the LLM groups
    run side by side on one GPU, each on its own inputs.  The LLM's prompts are the fixed seeded ids of SURVEY.md 8d, NOT built from this
    step's detections / transcripts — the streams share the GPU, not data (the multi-GPU placements add the hand-over explicitly:
    trackiellm_amd/dist.py PerceptionExchange)."""

    def __init__(self, tk, model, G, B, P, N, fused, rank, device, perception_batch, asr_steps, vision_device=None, audio_device=None):
        self.tk, self.G, self.B, self.P, self.N, self.fused = tk, G, B, P, N, fused
        hp = model.hparams
        self.sessions = [tk.LlmSession(model, B, P + N + 8) for _ in range(G)] if model is not None else []
        self.prompts = []
        for g in range(G):
            pr = np.stack([splitmix_tokens(3 + 1000 * ((rank * G + g) * B + s), P, 3, hp.vocab) for s in range(B)])
            pr[:, 0] = 1  # BOS
            self.prompts.append(pr)
        self.perc_ms = {"vision": [], "audio": []}
        self.n_dets = 0
        self.asr_steps = asr_steps
        if fused:
            self.PB = PB = max(1, min(G * B, perception_batch))  # frames / utterances per detector / ASR call
            self.det = tk.ObjectDetector(model="synthetic://yolov8n?seed=5&cls_bias=-0.45", width=640, height=640, conf=0.5, iou=0.5,
                                         device=device if vision_device is None else vision_device, max_batch=PB)
            self.asr = tk.Asr(hp=tk.WHISPER_TINY_EN(), seed=6, device=device if audio_device is None else audio_device, max_batch=PB)
            if os.environ.get("TK_BENCH_FAST_PERCEPTION") == "1":  # --fast-perception: the opt-in split-f16 contraction (never the default: parity is claimed on the exact path)
                self.det.set_fast_contraction(True)
                self.asr.set_fast_contraction(True)
            self.vad = tk.Vad()
            frng = np.random.default_rng(1 + rank)
            self.frames = [frng.integers(0, 256, (640, 640, 3), dtype=np.uint8) for _ in range(G * B)]
            prng = np.random.default_rng(2 + rank)
            self.pcm = np.clip(prng.normal(0, 3000, (G * B, 16000)), -32768, 32767).astype(np.int16)

    def _vision_pass(self):  # one frame per concurrent cycle, PB frames per detector call
        t = time.time()
        n = 0
        for i in range(0, self.G * self.B, self.PB):
            n += sum(len(r) for r in self.det.detect_batch(self.frames[i:i + self.PB]))
        self.n_dets = n
        self.perc_ms["vision"].append(1000 * (time.time() - t))

    def _audio_pass(self):  # one second of PCM per concurrent cycle
        t = time.time()
        for b in range(self.G * self.B):
            self.vad.reset()
            self.vad.process_with_events(self.pcm[b])
        for i in range(0, self.G * self.B, self.PB):
            self.asr.transcribe_tokens(self.pcm[i:i + self.PB], self.asr_steps, want_aux=False)
        self.perc_ms["audio"].append(1000 * (time.time() - t))

    def perception_async(self):
        th = [threading.Thread(target=self._vision_pass), threading.Thread(target=self._audio_pass)]
        for t in th:
            t.start()
        return th

    def prime(self):
        if self.fused:
            for t in self.perception_async():  # perception of the first timed batch
                t.join()
            self.perc_ms["vision"].clear(); self.perc_ms["audio"].clear()

    def step(self):
        th = self.perception_async() if self.fused else []
        res = [None] * self.G

        def llm_group(g):
            t_a = time.time()
            self.sessions[g].prefill(self.prompts[g])
            t_b = time.time()
            toks, ms_step = self.sessions[g].decode(self.B, self.N)
            res[g] = (toks, t_b - t_a, time.time() - t_b, ms_step)

        lt = [threading.Thread(target=llm_group, args=(g,)) for g in range(1, self.G)]
        for t in lt:
            t.start()
        if self.G:
            llm_group(0)
        for t in lt + th:
            t.join()
        if not self.G:
            return None, 0.0, 0.0, 0.0
        return res[0][0], float(np.mean([r[1] for r in res])), float(np.mean([r[2] for r in res])), float(np.mean([r[3] for r in res]))

    def run(self, steps, warmup, barrier=lambda: None, exchange=None):
        """exchange (model-per-gpu placement, LLM rank): every step ends with the hand-over of the perception ranks' results, and a step
        only generates once the results of ITS batch (handed over at the end of the previous step) are here"""
        self.prime()
        need = self.G * self.B
        for i in range(warmup):
            if exchange is not None and i > 0:
                exchange.require(need, need)
            self.step()
            if exchange is not None:
                exchange.hand_over(None)
        self.perc_ms["vision"].clear(); self.perc_ms["audio"].clear()
        barrier()
        t0 = time.time()
        pre_s = dec_s = 0.0
        ms_steps = []
        for i in range(steps):
            if exchange is not None and (warmup > 0 or i > 0):
                exchange.require(need, need)
            _, a, b, ms = self.step()
            pre_s += a
            dec_s += b
            ms_steps.append(ms)
            if exchange is not None:
                exchange.hand_over(None)
        barrier()
        elapsed = time.time() - t0
        return {"elapsed": elapsed, "prefill_s": pre_s / max(steps, 1), "decode_s": dec_s / max(steps, 1), "decode_ms_per_step": float(np.mean(ms_steps)) if ms_steps else 0.0}

    def close(self):
        for s in self.sessions:
            s.close()
        if self.fused:
            self.det.close(); self.asr.close(); self.vad.close()


def weight_bytes_per_shape(hp):
    """WEIGHT bytes one launch of each W4A8 shape streams (SURVEY.md §8d's per-token figure, split by launch)"""
    D, FF, V = hp.d_model, hp.d_ff, hp.vocab
    qd, kvd = hp.n_head * hp.head_dim, hp.n_kv_head * hp.head_dim
    return {"gate_up": 2 * FF * D * Q4K_BPW, "down_q6": D * FF * Q6K_BPW, "down_q4": D * FF * Q4K_BPW,
            "qkv_q6": (qd + kvd) * D * Q4K_BPW + kvd * D * Q6K_BPW, "qkv_q4": (qd + 2 * kvd) * D * Q4K_BPW, "o": D * qd * Q4K_BPW, "lm_head": V * D * Q6K_BPW}


def w_step_of(hp):
    """weights (= int8 multiply-adds per row) of one decode step"""
    qd, kvd = hp.n_head * hp.head_dim, hp.n_kv_head * hp.head_dim
    return hp.n_layer * (hp.d_model * (qd + 2 * kvd) + qd * hp.d_model + 3 * hp.d_model * hp.d_ff) + hp.vocab * hp.d_model


def gemv_roofline(sess, hp, rows, model_weight_bytes, iters=50):
    """W4A8 launch set of one decode step at `rows` rows per pass.  frac = WEIGHT bytes of the set / sum of launch durations / 8 TB/s —
    SURVEY.md §8d's bytes; the activation images and fp32 K-split partial outputs the launches also move are reported separately."""
    n_layer = hp.n_layer
    q6 = q6_layers(n_layer)
    q4 = [l for l in range(n_layer) if l not in q6]
    wb = weight_bytes_per_shape(hp)
    shapes, total_ms, total_w, total_all, launches = {}, 0.0, 0.0, 0.0, 0
    for name, which, layers in (("gate_up", 0, list(range(n_layer))), ("down_q6", 1, q6), ("down_q4", 1, q4), ("qkv_q6", 2, q6), ("qkv_q4", 2, q4),
                                ("o", 4, list(range(n_layer))), ("lm_head", 3, [0])):
        if not layers:
            continue
        ms, all_bytes = sess.time_gemv(layers[0], which, rows, iters)
        cnt = len(layers)
        shapes[name] = {"ms": round(ms, 5), "weight_GBps": round(wb[name] / ms / 1e6, 1), "launches_per_step": cnt}
        total_ms += ms * cnt
        total_w += wb[name] * cnt
        total_all += all_bytes * cnt
        launches += cnt
    assert abs(total_w - model_weight_bytes) < 1e-6 * model_weight_bytes, (total_w, model_weight_bytes)  # = the bytes a decode step streams
    achieved = total_w / total_ms / 1e6
    # > 32 rows: the K-streamed batched variants of the same arithmetic (16x16x64 MFMAs up to 128 rows, 32x32x32 above)
    kernel = "k_gemm32_w4a8" if rows > 128 else "k_gemm_w4a8" if rows > 32 else "k_gemv_w4a8"
    qd, kvd = hp.n_head * hp.head_dim, hp.n_kv_head * hp.head_dim
    w_step = hp.n_layer * (hp.d_model * (qd + 2 * kvd) + qd * hp.d_model + 3 * hp.d_model * hp.d_ff) + hp.vocab * hp.d_model
    tops = 2.0 * rows * w_step / (total_ms * 1e-3) / 1e12
    # which roof: a launch does 2 * rows int8 ops per weight and streams 0.5625 - 0.82 B per weight; the machine balance is 5000 TOP/s / 8 TB/s =
    # 625 op/B, so passes of more than ~190 rows are bound by the matrix cores (the 256-row headline), narrower ones by HBM
    ops_per_byte = 2.0 * rows * w_step_of(hp) / total_w
    mfma_bound = ops_per_byte > INT8_PEAK_TOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
    tops_now = 2.0 * rows * w_step_of(hp) / (total_ms * 1e-3) / 1e12
    out = {"bound": "mfma" if mfma_bound else "hbm", "kernel": kernel, "rows_per_pass": rows,
           "achieved": round(tops_now, 1) if mfma_bound else round(achieved, 1), "peak": INT8_PEAK_TOPS if mfma_bound else HBM_PEAK_GBS,
           "unit": "TOP/s (int8, dense)" if mfma_bound else "GB/s",
           "frac": round(tops_now / INT8_PEAK_TOPS, 4) if mfma_bound else round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "waste_ratio": None,
           "hbm_view": {"achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "what": "weight bytes of the launch set / sum of launch durations (SURVEY.md 8d's formula)"},
           "algorithmic_bytes_per_launch": round(total_w / launches), "avg_launch_ms": round(total_ms / launches, 5),
           "launches_per_decode_step": launches, "per_shape": shapes,
           "with_activations_and_partials": {"bytes_per_launch": round(total_all / launches), "achieved": round(total_all / total_ms / 1e6, 1),
                                             "frac": round(total_all / total_ms / 1e6 / HBM_PEAK_GBS, 4)},
           "int8_tops": round(tops, 1), "int8_peak_tops": INT8_PEAK_TOPS, "int8_frac": round(tops / INT8_PEAK_TOPS, 4),
           "int8_ops_per_weight_byte": round(2.0 * rows * w_step / total_w, 1)}
    # HBM traffic per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE collected separately, gfx950 correction applied): only a
    # summary collected from THIS kernel source counts; anything else would be a stale number
    pmc = newest_pmc("%s_b%d" % ("gemm" if rows > 32 else "gemv", rows))
    if os.path.exists(pmc):
        pj = json.load(open(pmc))
        if pj.get("rows_per_pass") == rows and pj.get("kernel_source_sha") == kernel_source_sha():
            out["traffic"] = pj.get("hbm_bytes_per_average_launch")
            out["waste_ratio"] = round(out["traffic"] / out["algorithmic_bytes_per_launch"], 3)
            out["traffic_source"] = os.path.relpath(pmc, ROOT)
    return out


def attention_roofline(sess, hp, rows, ctx, iters=64):
    ms, kvb = sess.time_attention(rows, ctx, iters)
    ach = kvb / ms / 1e6
    out = {"bound": "hbm", "kernel": "k_attention", "rows_per_pass": rows, "cached_positions": ctx, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": round(kvb), "avg_launch_ms": round(ms, 5),
           "launches_per_decode_step": hp.n_layer, "traffic": None, "waste_ratio": None}
    pmc = newest_pmc("attention_b%d" % rows)  # tools/pmc_gemv.py <fetch> <write> <rows> <out> k_attention <ctx>
    if os.path.exists(pmc):
        pj = json.load(open(pmc))
        if pj.get("rows_per_pass") == rows and pj.get("cached_positions") == ctx and pj.get("kernel_source_sha") == kernel_source_sha():
            out["traffic"] = pj.get("hbm_bytes_per_average_launch")
            out["waste_ratio"] = round(out["traffic"] / out["algorithmic_bytes_per_launch"], 3)
            out["traffic_source"] = os.path.relpath(pmc, ROOT)
    return out


def omp_threads(n):
    import ctypes
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass


def cpu_baseline(tk, model, hp, P, N, asr_steps):
    """One whole fused cycle on the CPU oracle ('port': this repo's restatement, not llama.cpp / ONNX Runtime / whisper.cpp, which cannot
    be built here — SURVEY.md §8d): detector 640x640 (preprocess + network + NMS), Whisper tiny.en (log-mel + encoder + asr_steps greedy
    steps), Mistral-7B Q4_K_M batched prompt prefill + greedy decode; timed at the box's core share and at 1 core on a bounded sample
    (decode tokens are timed for a few steps and extrapolated to N; everything else is timed whole)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, int(os.environ.get("TK_BENCH_CPU_CORES", "16")))  # the GPU box grants a 16-core share per GPU
    omp_threads(cores)
    cfg = O.LlmConfig(n_layer=hp.n_layer, d_model=hp.d_model, n_head=hp.n_head, n_kv_head=hp.n_kv_head, head_dim=hp.head_dim,
                      d_ff=hp.d_ff, vocab=hp.vocab, max_ctx=P + 16, max_seq=1, rms_eps=hp.rms_eps, rope_theta=hp.rope_theta,
                      ks_qkv=hp.ks_qkv, ks_o=hp.ks_o, ks_gateup=hp.ks_gateup, ks_down=hp.ks_down, ks_out=hp.ks_out)
    t0 = time.time()
    orc = O.OracleLlm(cfg, seed=4)
    yolo = O.OracleYolo(nc=80, seed=5, cls_bias=-0.45)
    wh = O.OracleWhisper(O.whisper_tiny_en(), seed=6)
    t_synth = time.time() - t0
    frame = np.random.default_rng(1).integers(0, 256, (640, 640, 3), dtype=np.uint8)
    pcm = np.clip(np.random.default_rng(2).normal(0, 3000, (1, 16000)), -32768, 32767).astype(np.int16)
    prompt = splitmix_tokens(3, P, 3, hp.vocab)
    prompt[0] = 1

    def one(n_cores, n_prompt, n_dec):
        omp_threads(n_cores)
        t = {}
        a = time.time()
        x = O.preprocess(frame, 640, 640, nhwc=True)[None]
        raw = yolo.forward(x)[0]
        dets = yolo.post(raw, 640, 640, 0.5, 0.5)
        t["detector_s"] = time.time() - a
        a = time.time()
        wt, _, _, _ = wh.transcribe(pcm, asr_steps)
        t["asr_s"] = time.time() - a
        orc.reset()
        a = time.time()
        _, am = orc.forward(np.zeros(n_prompt, np.int32), np.arange(n_prompt, dtype=np.int32), prompt[:n_prompt], want_logits=False)
        t["prefill_s_per_token"] = (time.time() - a) / n_prompt
        cur, toks = int(am[-1]), []
        a = time.time()
        for i in range(n_dec):
            toks.append(cur)
            _, am = orc.forward([0], [n_prompt + i], [cur], want_logits=False)
            cur = int(am[0])
        t["decode_s_per_token"] = (time.time() - a) / n_dec
        t["cycle_s"] = t["detector_s"] + t["asr_s"] + P * t["prefill_s_per_token"] + N * t["decode_s_per_token"]
        return t, toks, len(dets[0]), wt[0].tolist()

    full, toks, n_det, asr_ids = one(cores, P, 8)
    single, _, _, _ = one(1, 4, 2)  # 1 core: 4 prompt tokens + 2 decode tokens timed (a 7B token is ~16x slower on one core)
    omp_threads(cores)
    # parity spot-check against the GPU on the same weights and prompt
    chk = tk.LlmSession(model, 1, P + 16)
    first = chk.prefill(prompt[None, :])
    gt, _ = chk.decode(1, 8)
    chk.close()
    gtoks = [int(first[0])] + [int(v) for v in gt[:7, 0]]
    orc.close()
    r = lambda d: {k: round(v, 5) for k, v in d.items()}
    return {"value": round(1.0 / full["cycle_s"], 5), "unit": "cycles/s", "cores": cores, "kind": "port",
            "sample": f"EXTRAPOLATED: 8 of the {N} decode tokens timed and scaled to {N}; the rest timed whole.  ONE cycle on the oracle (CPU restatement, "
                      f"not llama.cpp/ORT/whisper.cpp): detector 640x640 + Whisper tiny.en mel/encoder/{asr_steps} steps + Mistral-7B Q4_K_M {P}-token batched "
                      f"prefill + {N}-token decode; weights synthesised in {t_synth:.0f} s (untimed)",
            "breakdown": r(full), "llm_tok_per_s": round(1.0 / full["decode_s_per_token"], 3),
            "one_core": {"value": round(1.0 / single["cycle_s"], 6), "unit": "cycles/s", "cores": 1,
                         "sample": "EXTRAPOLATED: 4 prompt + 2 decode tokens timed and scaled; the same cycle at 1 thread, detector + ASR timed whole",
                         "breakdown": r(single)},
            "token_ids_match_gpu": gtoks == toks, "detections": n_det, "asr_ids_head": asr_ids[:4]}


def reference_abi_b1(tk, hp, N, roof1):
    """What a drop-in host gets today at batch 1 through the reference's own entry points only: tk_model_loader_load_model ->
    tk_llm_runner_prepare_generation -> N x tk_llm_runner_generate_next_token (one sequence, one host round trip per token), and one
    tk_cortex_* cycle (frame -> detect -> prompt -> LLM response)."""
    out = {}
    loader = tk.ModelLoader()
    h = loader.load("synthetic://mistral-7b?seed=4")  # find-or-load: a second copy of the weights in HBM for the time of this leg
    runner = tk.LlmRunner(h, context_size=512)
    prompt = "x" * 63
    t0 = time.time()
    runner.prepare(prompt)
    t_pre = time.time() - t0
    for _ in range(8):  # warm
        runner.next_token()
    t0 = time.time()
    n = 0
    for _ in range(N):
        if runner.next_token() is None:
            break
        n += 1
    dt = time.time() - t0
    out["llm_runner"] = {"tokens": n, "tok_per_s": round(n / dt, 2), "ms_per_token": round(1000 * dt / max(n, 1), 4), "prefill_64_tokens_s": round(t_pre, 4),
                         "cycles_per_s_llm_only": round(1.0 / (t_pre + N * dt / max(n, 1)), 3),
                         "hbm_frac_end_to_end": round(hp_weight_bytes(tk, h) * n / dt / (HBM_PEAK_GBS * 1e9), 4)}
    runner.close()
    loader.unload(h)
    loader.close()
    out["k_gemv_w4a8_1_row"] = {k: roof1[k] for k in ("rows_per_pass", "achieved", "frac", "avg_launch_ms")}
    try:
        cx = tk.Cortex(llm="synthetic://mistral-7b?seed=4", detector="synthetic://yolov8n?seed=5&cls_bias=-0.45",
                       asr="synthetic://whisper-tiny.en?seed=6", vad="synthetic://vad?seed=7")
        cx.set_max_tokens(N)
        cx.start()
        frame = np.random.default_rng(1).integers(0, 256, (640, 640, 3), dtype=np.uint8)
        pcm = np.clip(np.random.default_rng(2).normal(0, 3000, 16000), -32768, 32767).astype(np.int16)
        done0 = cx.stats().llm_responses
        t0 = time.time()
        cx.inject_audio(pcm)
        cx.inject_frame(frame)
        while cx.stats().llm_responses == done0 and time.time() - t0 < 60:
            time.sleep(0.002)
        dt = time.time() - t0
        st = cx.stats()
        cx.stop()
        out["cortex"] = {"cycle_s": round(dt, 4), "cycles_per_s": round(1.0 / dt, 3), "llm_tokens": int(st.llm_tokens), "frames_with_objects": int(st.frames_with_objects)}
        cx.close()
    except Exception as e:  # reported, never fatal for the headline
        out["cortex"] = {"error": str(e)[:200]}
    return out


def prompt_processing(tk, model, lengths=(64, 448, 2048)):
    """one prompt of n tokens through TkLlmSession::prefill (passes of up to 256 rows; from position 128 on the attention of a pass runs 16 rows
    of the sequence per workgroup on the fp32 matrix pipe, csrc/llm/tk_llm_kernels.hip: k_attention_prefill) — the reference feeds whole prompts
    (src/ai_models/tk_runner_streaming.c:20-40) and budgets 2 048 tokens for them (src/cortex/tk_cortex_main.c:1334).  Best of two timed runs
    after the capture run; ids of both runs equal."""
    hp = model.hparams
    rng = np.random.default_rng(5)
    out = []
    for n in lengths:
        sess = tk.LlmSession(model, 1, n + 8)
        toks = rng.integers(3, hp.vocab, (1, n)).astype(np.int32)
        first = sess.prefill(toks)
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            again = sess.prefill(toks)
            best = min(best, time.perf_counter() - t0)
            if not np.array_equal(first, again):
                raise RuntimeError("prompt processing is not reproducible")
        out.append({"prompt_tokens": n, "ms": round(1e3 * best, 2), "tok_per_s": round(n / best, 1)})
        sess.close()
    return out


def long_context_decode(tk, model, contexts=(2000, 3900), steps=64):
    """one sequence with a `ctx`-token prompt, then 2 x `steps` greedy tokens through the device-side loop (the second `steps` timed): decode at
    the context lengths the reference budgets for (2 048-token prompts in a 4 096-position window: src/cortex/tk_cortex_main.c:1334,
    src/ai_models/tk_runner_lifecycle.c:48).  From position 640 on a one-row pass runs its attention as scores + PV chains spread over the chip
    (csrc/llm/tk_llm_kernels.hip: k_att_scores_long / k_att_pv_chain) instead of one latency chain per pair of heads."""
    hp = model.hparams
    rng = np.random.default_rng(9)
    out = []
    for ctx in contexts:
        sess = tk.LlmSession(model, 1, ctx + 2 * steps + 8)
        sess.prefill(rng.integers(3, hp.vocab, (1, ctx)).astype(np.int32))
        sess.decode(1, steps)
        _, ms = sess.decode(1, steps)
        out.append({"positions": [ctx + steps, ctx + 2 * steps], "ms_per_token": round(ms, 3), "tok_per_s": round(1000.0 / ms, 1)})
        sess.close()
    return out


BATCHED_CORTEX_DEADLINE_S = 120
STALLED = []  # legs that hit their deadline: main() then exits without running destructors


def reference_abi_batched_cortex(tk, K, N, progress=False):
    """K cortex handles (tk_cortex_create ... tk_cortex_destroy only) that share one LLM model file, each driven by its own host thread through
    ONE data-dependent cycle of the reference's loop (/root/reference/src/cortex/tk_cortex_main.c:1149-1237, 1323-1379): 1 s of PCM + the
    silence that ends the segment -> VAD -> Whisper -> the transcript becomes a conversation turn -> reasoner context string -> LLM response;
    then a 640 x 640 frame -> detector -> the detections enter the context string -> LLM response.  The prompts are built from THAT
    cycle's transcript and detections (different frames / audio per cortex); the LLM rows of all cortices are decoded together behind the
    runner API (csrc/llm/tk_llm_batcher.h).  Two responses of N / 2 tokens each = N tokens per cycle.  Perception is called per handle, one frame /
    utterance per call as the reference's API shapes it; the handles' calls are coalesced by the per-model-file engines behind it (round 6).  The
    synthetic vocabulary is byte-level, so a context string costs one prompt token per byte."""
    # the runners the cortices create share ONE decode session of K sequence slots (the library's default is 16 per session: 256 cortices would
    # decode in sixteen separate 16-row streams); read when the first runner of the model is created
    prev_slots = os.environ.get("TK_MI355X_RUNNER_SLOTS")
    os.environ["TK_MI355X_RUNNER_SLOTS"] = os.environ.get("TK_BENCH_CORTEX_SLOTS") or str(min(K, 256))
    t_create = time.time()
    cxs = [tk.Cortex(llm="synthetic://mistral-7b?seed=4", detector="synthetic://yolov8n?seed=5&cls_bias=-0.45", asr="synthetic://whisper-tiny.en?seed=6",
                     vad="synthetic://vad?seed=7") for _ in range(K)]
    t_create = time.time() - t_create
    if prev_slots is None:
        os.environ.pop("TK_MI355X_RUNNER_SLOTS", None)
    else:
        os.environ["TK_MI355X_RUNNER_SLOTS"] = prev_slots
    for cx in cxs:
        cx.set_max_tokens(max(N // 2, 1))
        cx.start()
    done = [0.0] * K
    plen = [0] * K

    def drive(i):
        rng = np.random.default_rng(1000 + i)
        frame = rng.integers(0, 256, (640, 640, 3), dtype=np.uint8)
        pcm = np.concatenate([np.clip(rng.normal(0, 9000, 16000), -32768, 32767), np.zeros(9600)]).astype(np.int16)
        cx = cxs[i]
        base = cx.stats().llm_responses
        for chunk in np.split(pcm, 16):  # 100 ms chunks, like the reference's microphone worker
            cx.inject_audio(chunk)
        want = base + (1 if cx.stats().speech_segments > 0 else 0)
        t_end = time.time() + 600
        while cx.stats().llm_responses < want and time.time() < t_end:
            time.sleep(0.002)
        dropped = cx.stats().events_dropped
        cx.inject_frame(frame)
        while cx.stats().frames_processed < 1 and cx.stats().events_dropped == dropped and time.time() < t_end:  # (a frame the pipeline failed on is counted as dropped)
            time.sleep(0.002)
        want += 1 if cx.stats().frames_with_objects > 0 else 0
        while cx.stats().llm_responses < want and time.time() < t_end:
            time.sleep(0.002)
        done[i] = time.time()
        plen[i] = len(cx.last_prompt())

    th = [threading.Thread(target=drive, args=(i,), daemon=True) for i in range(K)]
    t0 = time.time()
    ticking = [progress]

    def ticker():  # developer aid (tools/time_batched_cortex.py --progress): where the K cycles stand, every few seconds, on stderr
        while ticking[0]:
            time.sleep(5.0)
            st = [cx.stats() for cx in cxs]
            print("[%6.1f s] speech segments %d, frames processed %d (with objects %d), responses %d, tokens %d, dropped events %d" % (
                time.time() - t0, sum(s.speech_segments for s in st), sum(s.frames_processed for s in st), sum(s.frames_with_objects for s in st),
                sum(s.llm_responses for s in st), sum(s.llm_tokens for s in st), sum(s.events_dropped for s in st)), file=sys.stderr, flush=True)

    tick = threading.Thread(target=ticker, daemon=True)
    tick.start()
    for t in th:
        t.start()
    # an extra must never hang the bench: K cycles take seconds; after BATCHED_CORTEX_DEADLINE_S the leg is reported as an error, its handles are
    # left alone (closing a wedged cortex could block) and main() leaves through os._exit once the JSON line is out
    deadline = time.time() + BATCHED_CORTEX_DEADLINE_S
    for t in th:
        t.join(max(0.0, deadline - time.time()))
    ticking[0] = False
    if any(t.is_alive() for t in th):
        STALLED.append("reference_abi_batched_cortex(%d)" % K)
        st = [cx.stats() for cx in cxs]
        raise RuntimeError("stalled: after %d s only %d of %d cycles were complete (responses %d, frames %d, speech segments %d)" % (
            BATCHED_CORTEX_DEADLINE_S, sum(1 for d_ in done if d_ > 0), K, sum(s_.llm_responses for s_ in st), sum(s_.frames_processed for s_ in st),
            sum(s_.speech_segments for s_ in st)))
    dt = max(done) - t0
    st = [cx.stats() for cx in cxs]
    # the shared model's scheduler counters (the registry hands this loader the cortices' own model): how wide the LLM passes really were
    ld = tk.ModelLoader()
    hm = ld.load("synthetic://mistral-7b?seed=4")
    passes, rows, widest = tk.ModelLoader.batch_stats(hm)
    ld.unload(hm)
    ld.close()
    # the shared perception engines' counters (round 6: every cortex's detector / ASR handle on one model file shares one batched engine whose
    # scheduler coalesces the handles' one-frame / one-utterance calls): a probe handle on the same file reads them
    dprobe = tk.ObjectDetector(model="synthetic://yolov8n?seed=5&cls_bias=-0.45", conf=0.5, iou=0.5)
    aprobe = tk.Asr(model="synthetic://whisper-tiny.en?seed=6")
    dshare, ashare = dprobe.share_stats(), aprobe.share_stats()
    dprobe.close()
    aprobe.close()
    for cx in cxs:
        cx.stop()
    for cx in cxs:
        cx.close()
    return {"cortices": K, "host_threads": K, "wall_s": round(dt, 3), "cycles_per_s": round(K / dt, 3), "create_s": round(t_create, 2),
            "shared_perception": {"detector_jobs": dshare[1], "frames": dshare[2], "widest_detector_job": dshare[3], "asr_jobs": ashare[1], "utterances": ashare[2],
                                  "widest_asr_job": ashare[3]},
            "llm_responses": int(sum(s.llm_responses for s in st)), "llm_tokens": int(sum(s.llm_tokens for s in st)),
            "speech_segments": int(sum(s.speech_segments for s in st)), "frames_with_objects": int(sum(s.frames_with_objects for s in st)),
            "last_prompt_bytes_mean": int(np.mean(plen)), "llm_passes": int(passes), "llm_rows_per_pass": round(rows / max(passes, 1), 1), "widest_pass": int(widest),
            "data_dependent": True}


def perception_fast_contraction(tk, device, asr_steps):
    """VERDICT r05 item 8's second `perception` object: the opt-in fast contraction (convolutions / the ASR's long passes on the f16 matrix pipe with
    split operands, tk_mi355x_detector_set_fast_contraction / tk_mi355x_asr_set_fast_contraction) beside the exact path, stand-alone, 32 frames and
    32 one-second clips per call; the headline and every parity claim stay on the exact path."""
    PB = 32
    rng = np.random.default_rng(1)
    frames = [rng.integers(0, 256, (640, 640, 3), dtype=np.uint8) for _ in range(PB)]
    pcm = np.clip(np.random.default_rng(2).normal(0, 3000, (PB, 16000)), -32768, 32767).astype(np.int16)

    def ms(fn, n=3):
        fn()
        t = time.time()
        for _ in range(n):
            fn()
        return round(1000.0 * (time.time() - t) / n, 2)

    det = tk.ObjectDetector(model="synthetic://yolov8n?seed=5&cls_bias=-0.45", width=640, height=640, conf=0.5, iou=0.5, device=device, max_batch=PB)
    x = rng.random((1, 640, 640, 3), dtype=np.float32)
    d_exact, raw_e, dets_e = ms(lambda: det.detect_batch(frames)), det.forward_raw(x), det.detect_batch(frames)
    det.set_fast_contraction(True)
    d_fast, raw_f, dets_f = ms(lambda: det.detect_batch(frames)), det.forward_raw(x), det.detect_batch(frames)
    det.close()
    asr = tk.Asr(hp=tk.WHISPER_TINY_EN(), seed=6, device=device, max_batch=PB)
    a_exact = ms(lambda: asr.transcribe_tokens(pcm, asr_steps, want_aux=False))
    tok_e, _, enc_e, _ = asr.transcribe_tokens(pcm[:2], asr_steps, want_aux=True)
    asr.set_fast_contraction(True)
    a_fast = ms(lambda: asr.transcribe_tokens(pcm, asr_steps, want_aux=False))
    tok_f, _, enc_f, _ = asr.transcribe_tokens(pcm[:2], asr_steps, want_aux=True)
    asr.close()
    same_sets = all(sorted((a[0], a[3]) for a in e) == sorted((b[0], b[3]) for b in f) for e, f in zip(dets_e, dets_f))
    return {"what": "opt-in split-f16 contraction beside the exact fp32 chains, stand-alone, per 32 frames / 32 one-second clips (not the headline's path)",
            "dtype": "f16 x 2 halves per operand on v_mfma_f32_32x32x16_f16, fp32 accumulation",
            "detector_ms_per_32": {"exact": d_exact, "fast": d_fast}, "asr_ms_per_32": {"exact": a_exact, "fast": a_fast, "forced_steps": asr_steps},
            "head_maps_max_err_of_scale": float(np.abs(raw_f - raw_e).max() / np.abs(raw_e).max()),
            "encoder_max_err_of_scale": float(np.abs(enc_f - enc_e).max() / np.abs(enc_e).max()),
            "detections_same_set_per_frame": bool(same_sets), "asr_ids_equal": bool(np.array_equal(tok_f, tok_e))}


def reference_abi_runners(tk, K, N):
    """K tk_llm_runner_t handles on ONE model handle, each driven by its own host thread through tk_llm_runner_prepare_generation /
    tk_llm_runner_generate_next_token only: what a host gets when it opens K runners instead of one (continuous batching behind the
    reference ABI, csrc/llm/tk_llm_batcher.h).  64-token prompts, N tokens each; LLM stream only."""
    loader = tk.ModelLoader()
    h = loader.load("synthetic://mistral-7b?seed=4")
    tk.ModelLoader.set_runner_slots(h, K)
    runners = [tk.LlmRunner(h, context_size=256) for _ in range(K)]
    prompts = ["".join(chr(97 + (i * 7 + j) % 26) for j in range(63)) for i in range(K)]
    counts = [0] * K

    def drive(i):
        runners[i].prepare(prompts[i])
        n = 0
        for _ in range(N):
            if runners[i].next_token() is None:
                break
            n += 1
        counts[i] = n

    for rnd in range(2):  # the first round captures the pass graphs
        p0, r0, _ = tk.ModelLoader.batch_stats(h)
        th = [threading.Thread(target=drive, args=(i,)) for i in range(K)]
        t0 = time.time()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.time() - t0
    p1, r1, widest = tk.ModelLoader.batch_stats(h)
    for r in runners:
        r.close()
    loader.unload(h)
    loader.close()
    return {"runners": K, "host_threads": K, "tokens_per_runner": int(np.mean(counts)), "wall_s": round(dt, 4), "cycles_per_s_llm_only": round(K / dt, 3),
            "tok_per_s": round(sum(counts) / dt, 1), "passes": int(p1 - p0), "rows": int(r1 - r0), "rows_per_pass": round((r1 - r0) / max(p1 - p0, 1), 2),
            "widest_pass": int(widest)}


def reference_abi_runners_c_host(K, N):
    """the same leg driven by a C program (tools/abi_runners_host.c: pthreads, no interpreter): what part of the Python leg's loss against
    direct sessions is the driver.  Built in place with gcc (the GPU box runs this image); None when that fails."""
    import subprocess
    exe = os.path.join(ROOT, "build", "abi_runners_host")
    libdir = os.path.join(ROOT, "trackiellm_amd")
    try:
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "abi_runners_host.c"), "-L" + libdir,
                               "-ltrackie_mi355x", "-lpthread", "-Wl,-rpath," + libdir, "-o", exe], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = subprocess.run([exe, str(K), str(N)], capture_output=True, text=True, timeout=300)
        return json.loads(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 else {"error": (out.stderr or "rc %d" % out.returncode)[-200:]}
    except Exception as e:
        return {"error": str(e)[:200]}


def hp_weight_bytes(tk, handle):
    tk.lib().tk_mi355x_llm_model_weight_bytes.restype = __import__("ctypes").c_uint64
    return tk.lib().tk_mi355x_llm_model_weight_bytes(handle)


COLL_CUDA = True  # collectives on device tensors (RCCL); False in the one-GPU rehearsal (gloo)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="concurrent cortex cycles per decode group = rows per LLM pass (<=256: sixteen 16-row MFMA M-tiles)")
    ap.add_argument("--prompt", type=int, default=64)
    ap.add_argument("--decode", type=int, default=128)
    ap.add_argument("--layers", type=int, default=32, help="debug only: fewer layers => result marked invalid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only: skip north_star_point, reference_abi_b1 and cpu_baseline")
    ap.add_argument("--sessions", type=int, default=3,
                    help="independent decode groups of --batch cycles run concurrently on their own HIP streams (fills the "
                         "launch/ramp bubbles of one group with another group's kernels); concurrent cycles = sessions * batch")
    ap.add_argument("--llm-only", action="store_true", help="configs[1] only: leave the detector / ASR / VAD streams out (marked in config)")
    ap.add_argument("--pipeline", action="store_true",
                    help="opt-in: the ranks form ONE layer-sharded LLM pipeline (RCCL send / recv of the residual stream between consecutive "
                         "GPUs, SURVEY.md 8e) instead of independent replicas; LLM stream only; --sessions row groups keep the stages busy")
    ap.add_argument("--pipe-rccl", action="store_true", help="pipeline hand-off as ncclSend / ncclRecv of the fp32 stream (RCCL over xGMI) instead of the peer-mapped "
                    "mailboxes: needs one GPU per stage and fails otherwise")
    ap.add_argument("--pipe-f16", action="store_true", help="pipeline hand-off payload in IEEE f16 (SURVEY.md 8e's 8 KiB per row) instead of exact fp32")
    ap.add_argument("--placement", choices=["replicas", "model-per-gpu", "combined"], default="replicas",
                    help="model-per-gpu (SURVEY.md 8e): rank 0 runs the LLM for ALL cycles of the job, the other ranks run the detector / ASR / VAD "
                         "streams for them; results travel as bytes through the host (no data-path collective).  combined (BASELINE configs[4]): the "
                         "LLM layer-sharded over the first ranks, detector and VAD + ASR on the spare GPUs (D.combined_roles)")
    ap.add_argument("--roofline-only", action="store_true",
                    help="only the isolated per-shape timing of the dominant kernel (the roofline objects); profile THIS command with "
                         "rocprofv3 --kernel-trace to compare its kernel durations with the HIP-event numbers (tools/roofline_check.py)")
    ap.add_argument("--perception-batch", type=int, default=64, help="frames / utterances per detector / ASR call")
    ap.add_argument("--fast-perception", action="store_true", help="A/B only: detector and ASR on the opt-in fast contraction (split-f16 operands, ~1e-6 of scale off the exact chains); the line is marked and is not the headline")
    ap.add_argument("--asr-steps", type=int, default=16, help="forced greedy decoder steps per utterance (SURVEY.md 8d)")
    ap.add_argument("--ns-steps", type=int, default=4, help="timed steps of the north_star_point run")
    ap.add_argument("--cortices", type=lambda v: [int(x) for x in v.split(",") if x], default=[16, 64],
                    help="cortex-handle counts of the reference_abi_batched_cortex extras.  256 handles need a 4096-position KV cache for 256 sequences (137 GB) "
                         "beside what this process already holds: measured on its own by tools/time_batched_cortex.py 256 (profiles/r06_batched_cortex_256.txt: "
                         "33.7 cycles/s; the detector / ASR engines are shared per model file since round 6)")
    ap.add_argument("--weights", choices=["q4_k_m", "f16"], default="q4_k_m",
                    help="f16: BASELINE configs[4]'s fp16 checkpoint (14.2 GB of weights per decode step) on the exact fp32 MFMA GEMM; LLM stream only, "
                         "no W4A8 roofline object (other kernels)")
    args = ap.parse_args()
    if args.fast_perception:
        os.environ["TK_BENCH_FAST_PERCEPTION"] = "1"

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    from trackiellm_amd import dist as D
    # TK_BENCH_SHARE_GPU=1: a REHEARSAL of the N > 1 paths on a box with one GPU — every rank drives device 0, gloo lines the ranks up.
    # It exercises the launch contract, the handle exchange and the in-library stage hand-off across processes (hipIpc); the value it
    # prints says nothing about scaling and is marked so.
    share = os.environ.get("TK_BENCH_SHARE_GPU") == "1" and world > 1
    if world > 1:
        dist = D.init("gloo" if share else "nccl", local_rank)
    if share:
        local_rank = 0
    global COLL_CUDA
    COLL_CUDA = not share

    import trackiellm_amd as tk
    if tk.lib().tk_mi355x_device_count() <= local_rank:
        raise SystemExit("bench.py needs one MI355X per rank: no fallback path exists")
    tk.lib().tk_mi355x_set_default_device(local_rank)

    B, P, N = args.batch, args.prompt, args.decode
    G = max(1, args.sessions)
    if args.pipe_rccl:
        G = 1  # one communicator per device: the collective transport carries one row group (D.LibPipeline refuses more)
    hp = tk.MISTRAL_7B()
    hp.n_layer = args.layers

    def barrier():
        if dist is not None:
            D.barrier(dist, cuda=COLL_CUDA)

    if args.pipeline or (args.placement == "combined" and world > 1):
        return run_pipeline(args, tk, D, dist, hp, rank, local_rank, world, G, B, P, N, combined=args.placement == "combined" and world > 1)
    if args.placement == "model-per-gpu" and world > 1:
        return run_model_per_gpu(args, tk, D, dist, hp, rank, local_rank, world, G, B, P, N)

    t0 = time.time()
    f16 = args.weights == "f16"
    model = tk.LlmModel(hp, device=local_rank).fill_synthetic(4, f16=f16)
    hp = model.hparams
    t_load = time.time() - t0
    if f16:
        args.llm_only, args.no_extras = True, True
    if args.roofline_only:
        args.llm_only, args.steps, args.warmup = True, 0, 0
    fused = not args.llm_only
    cb = CycleBench(tk, model, G, B, P, N, fused, rank, local_rank, args.perception_batch, args.asr_steps)
    if os.environ.get("TK_BENCH_DUMP_MAPS"):  # the load map of THIS process, for symbolising a profiler abort against it (tools/symbolise_crash.py)
        with open("/proc/self/maps") as f, open(os.environ["TK_BENCH_DUMP_MAPS"], "w") as g:
            g.write(f.read())
    r = cb.run(args.steps, args.warmup, barrier) if args.steps > 0 else None
    elapsed = r["elapsed"] if r else 0.0
    if dist is not None and r:
        elapsed = D.max_over_ranks(dist, elapsed, cuda=COLL_CUDA)
    if rank != 0:
        cb.close()
        if dist is not None:
            dist.destroy_process_group()
        return

    sess = cb.sessions[0]
    if f16:  # other kernels (k_gemm_f32 on f16 weights): the whole-step view is the roofline statement of this run
        value = D.aggregate_throughput(G * B, args.steps, world, elapsed)
        dec_ms = r["decode_ms_per_step"]
        print(json.dumps({"metric": METRIC, "value": round(value, 3), "unit": "cycles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1000.0 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f16 weights x f16-rounded f32 activations, fp32 MFMA chain", "data": "synthetic",
                          "config": {"workload": "configs[4] weights on one GPU: Mistral-7B fp16, 64-token prefill + 128-token greedy decode per cycle (LLM stream only)",
                                     "concurrent_cycles_per_gpu": G * B, "decode_groups": G, "rows_per_llm_pass": B, "parallelism": f"replicas x{world}"},
                          "llm_tok_per_s": round(G * B * world * N / r["decode_s"], 1), "decode_ms_per_step": round(dec_ms, 4),
                          "weight_bytes_per_decode_step": int(model.weight_bytes),
                          "llm_decode_step_roofline": {"bytes_per_step": int(model.weight_bytes), "steps_per_s": round(G * 1000.0 / dec_ms, 1),
                                                       "frac": round(model.weight_bytes * (G * 1000.0 / dec_ms) / (HBM_PEAK_GBS * 1e9), 4)}}))
        cb.close()
        if dist is not None:
            dist.destroy_process_group()
        return
    roofline = gemv_roofline(sess, hp, B, model.weight_bytes)
    roofline_att = attention_roofline(sess, hp, B, min(P + N // 2, P + N))
    if args.roofline_only:
        r16 = gemv_roofline(sess, hp, 16, model.weight_bytes) if B != 16 else roofline
        print(json.dumps({"metric": METRIC, "value": None, "unit": "cycles/s", "n_gpus": world, "note": "roofline-only run: no timed steps",
                          "kernel_source_sha": kernel_source_sha(), "roofline": roofline, "roofline_attention": roofline_att,
                          "roofline_16_rows": {k: r16[k] for k in ("rows_per_pass", "achieved", "frac", "avg_launch_ms", "per_shape")}}))
        return
    value = D.aggregate_throughput(G * B, args.steps, world, elapsed)
    dec_ms = r["decode_ms_per_step"]
    kv_step = B * 131072 * (P + N / 2.0) * (hp.n_layer / 32.0)
    out = {
        "metric": METRIC, "value": round(value, 3), "unit": "cycles/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / args.steps, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": ("configs[3] fused cycle: YOLOv8n 640x640 frame (preprocess+network+NMS) + 1 s PCM (VAD + Whisper-tiny.en "
                                "log-mel/encoder/%d forced decoder steps) + Mistral-7B Q4_K_M 64-token prefill and 128-token greedy decode, "
                                "3 concurrent HIP streams" % args.asr_steps) if fused else
                               "configs[1]: Mistral-7B Q4_K_M, 64-token prefill + 128-token greedy decode per cycle (LLM stream only)",
                   "concurrent_cycles_per_gpu": G * B, "decode_groups": G, "rows_per_llm_pass": B, "prompt_tokens": P, "decode_tokens": N, "layers": hp.n_layer,
                   "k_split": [hp.ks_qkv, hp.ks_o, hp.ks_gateup, hp.ks_down], "parallelism": f"replicas x{world}"},
        "llm_tok_per_s": round(G * B * world * N / r["decode_s"], 1),
        "decode_ms_per_step": round(dec_ms, 4), "prefill_s_per_cycle_batch": round(r["prefill_s"], 4),
        "model_load_s": round(t_load, 2), "weight_bytes_per_decode_step": int(model.weight_bytes),
        "roofline": roofline, "roofline_attention": roofline_att,
        # SURVEY.md 8d's whole-step view: (weights + B * 128 KiB * mean context of KV per row) per decode step over 8 TB/s, all groups in flight
        "llm_decode_step_roofline": {"bytes_per_step": int(model.weight_bytes + kv_step), "steps_per_s": round(G * 1000.0 / dec_ms, 1),
                                     "frac": round((model.weight_bytes + kv_step) * (G * 1000.0 / dec_ms) / (HBM_PEAK_GBS * 1e9), 4)},
    }
    if fused:
        out["perception"] = {"vision_ms_per_batch": round(float(np.mean(cb.perc_ms["vision"])), 2), "audio_ms_per_batch": round(float(np.mean(cb.perc_ms["audio"])), 2),
                             "detections_last_batch": cb.n_dets, "overlapped_with_llm": True,
                             "dtype": "split-f16 operands on the f16 MFMA (--fast-perception: opt-in, ~1e-6 of scale off the exact chains)" if args.fast_perception else "f32 (exact fp32 MFMA chain)"}
        if args.fast_perception:
            out["not_the_headline"] = "--fast-perception: the perception streams ran the opt-in fast contraction; parity is claimed on the exact path only"
    if args.layers != 32:
        out["invalid"] = "debug run with fewer layers"
    if not COLL_CUDA:
        out["rehearsal"] = "TK_BENCH_SHARE_GPU=1: all ranks share ONE GPU; not a scaling measurement"
    extras = not args.no_extras and args.layers == 32 and world == 1
    roof16 = None
    if extras and not (G == 3 and B == 16):
        # ---- the configuration that meets north_star's two conditions together: 3 groups x 16 rows per pass (every group streams the
        # weights itself: k_gemv_w4a8 at one M-tile, the HBM-bound regime), same fused workload, its own short timed run ----
        cb.close()
        ns = CycleBench(tk, model, 3, 16, P, N, fused, rank, local_rank, args.perception_batch, args.asr_steps)
        nr = ns.run(args.ns_steps, 1)
        roof16 = gemv_roofline(ns.sessions[0], hp, 16, model.weight_bytes)
        ns_value = 48 * args.ns_steps / nr["elapsed"]
        n_pre = 3 * (-(-(16 * (P - 1)) // 256) + 1)  # prompt passes per step (256-row passes + the sampling pass), all groups
        passes = 3 * N + n_pre
        out["north_star_point"] = {
            "what": "north_star: >= 30 fused cycles/s on 1 MI355X at >= 40 % of the HBM roofline — both conditions in ONE timed run",
            "config": {"decode_groups": 3, "rows_per_llm_pass": 16, "concurrent_cycles": 48, "fused": fused}, "steps": args.ns_steps, "warmup": 1,
            "value": round(ns_value, 3), "unit": "cycles/s", "ms_per_step": round(1000.0 * nr["elapsed"] / args.ns_steps, 2),
            "roofline": {k: roof16[k] for k in ("bound", "kernel", "rows_per_pass", "achieved", "peak", "unit", "frac", "avg_launch_ms", "with_activations_and_partials", "traffic", "waste_ratio")},
            "decode_window_hbm_frac": round((model.weight_bytes + 16 * 131072 * (P + N / 2.0)) * (3 * 1000.0 / nr["decode_ms_per_step"]) / (HBM_PEAK_GBS * 1e9), 4),
            "whole_run_hbm_frac": round(model.weight_bytes * passes * args.ns_steps / nr["elapsed"] / (HBM_PEAK_GBS * 1e9), 4),
            "meets_30_cycles_per_s": bool(ns_value >= 30.0), "meets_40_percent_kernel_roofline": bool(roof16["frac"] >= 0.40),
            "meets_40_percent_whole_run": bool(model.weight_bytes * passes * args.ns_steps / nr["elapsed"] / (HBM_PEAK_GBS * 1e9) >= 0.40)}
        ns.close()
    else:
        cb.close()
    if extras:
        s1 = tk.LlmSession(model, 1, 64)
        roof1 = gemv_roofline(s1, hp, 1, model.weight_bytes)
        s1.close()
        out["reference_abi_b1"] = reference_abi_b1(tk, hp, N, roof1)
        out["prompt_processing"] = prompt_processing(tk, model)
        out["long_context_decode"] = long_context_decode(tk, model)
        out["reference_abi_batched"] = [reference_abi_runners(tk, K, N) for K in (16, 64, 256)]
        # the same runners from a C host: the difference is the Python driver (GIL hand-offs between K threads that each make one ctypes call per token)
        out["reference_abi_batched_c_host"] = [reference_abi_runners_c_host(K, N) for K in (16, 256)]
        # data-dependent cycles through tk_cortex_* only, K cortices on one model file (prompts built from each cycle's own detections / transcript)
        out["reference_abi_batched_cortex"] = []
        for K in args.cortices:
            try:
                out["reference_abi_batched_cortex"].append(reference_abi_batched_cortex(tk, K, N))
            except Exception as e:  # reported, never fatal for the headline
                out["reference_abi_batched_cortex"].append({"cortices": K, "error": str(e)[:200]})
    if extras and fused:
        try:
            out["perception_fast_contraction"] = perception_fast_contraction(tk, local_rank, args.asr_steps)
        except Exception as e:  # reported, never fatal for the headline
            out["perception_fast_contraction"] = {"error": str(e)[:200]}
    if extras and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(tk, model, hp, P, N, args.asr_steps)
    print(json.dumps(out), flush=True)
    if STALLED:  # a leg was abandoned with live handles: do not wait for their destructors
        os._exit(0)
    if dist is not None:
        dist.destroy_process_group()


def run_pipeline(args, tk, D, dist, hp, rank, local_rank, world, G, B, P, N, combined=False):
    """--pipeline: every rank is one LLM stage.  --placement combined (BASELINE configs[4]): D.combined_roles(world) — the first ranks
    form the pipeline, the spare GPUs run the detector and VAD + ASR for the same cycles.  The stage hand-off is the library's own
    (device mailboxes mapped with hipIpc, hipGraph replays, no host synchronisation per pass: csrc/llm/tk_llm_pipe.h); torch.distributed
    only carries the 80-byte mailbox handles once, the barrier and the max-over-ranks time."""
    f16 = args.weights == "f16"  # BASELINE configs[4]: the fp16 checkpoint layer-sharded over the node
    roles = D.combined_roles(world) if combined else {"llm": list(range(world)), "vision": [], "audio": []}
    if dist is None:
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist = D.init("gloo")
    cuda_t = world > 1 and COLL_CUDA
    cycles = G * B
    in_llm = rank in roles["llm"]
    model = pipe = cb = None
    if in_llm:
        model = tk.LlmModel(hp, device=local_rank).fill_synthetic(4, f16=f16)  # every stage builds the same weights and runs only its layers
        hp = model.hparams
        make = D.gpu_pipe_factory(tk, model, B, P + N + 8, payload_f16=args.pipe_f16)
        pipe = D.LibPipeline(dist, roles["llm"], hp.n_layer, G, make, rccl=args.pipe_rccl)
    else:
        D.LibPipeline(dist, roles["llm"], hp.n_layer, G, None, rccl=args.pipe_rccl)  # takes part in the handle exchange, owns no stage
        share_v = -(-cycles // len(roles["vision"])) if rank in roles["vision"] else 0
        share_a = -(-cycles // len(roles["audio"])) if rank in roles["audio"] else 0
        cb = PerceptionBench(tk, share_v, share_a, rank, local_rank, args.perception_batch, args.asr_steps)
    same = [np.stack([splitmix_tokens(3 + 1000 * (g * B + s_), P, 3, hp.vocab) for s_ in range(B)]) for g in range(G)]
    for pr in same:
        pr[:, 0] = 1

    ex = D.PerceptionExchange(dist, dst=roles["llm"][0], plan=D.perception_plan(roles["vision"], roles["audio"], cycles, args.asr_steps),
                              device="cuda" if cuda_t else None) if combined and world > 1 else None
    n_done = [0]

    def step():
        """one cycle batch: stage 0 needs the perception results of THIS batch (handed over at the end of the previous step) before it
        enqueues the batch's passes; every rank ends the step with the hand-over of the batch the perception ranks just finished"""
        res = None
        if in_llm:
            if ex is not None and rank == roles["llm"][0] and n_done[0] > 0:
                ex.require(cycles, cycles)
            pipe.generate(same, N)
        else:
            res = cb.step()
        if ex is not None:
            ex.hand_over(res)
        n_done[0] += 1

    for _ in range(args.warmup):
        step()
    D.barrier(dist, cuda=cuda_t)
    t0 = time.time()
    for _ in range(args.steps):
        step()
    D.barrier(dist, cuda=cuda_t)
    elapsed = D.max_over_ranks(dist, time.time() - t0, cuda=cuda_t)
    if rank == 0:
        n_st = len(roles["llm"])
        bounds = D.stage_bounds(hp.n_layer, n_st)
        par = "pipeline x%d (in-library hand-off of [rows, 4096] %s through peer-mapped device mailboxes, hipGraph replays)" % (n_st, "f16" if args.pipe_f16 else "fp32")
        if args.pipe_rccl:
            par = "pipeline x%d (in-library hand-off of [rows, 4096] fp32 by ncclSend / ncclRecv over RCCL, eager launches)" % n_st
        if combined:
            par += "; detector on ranks %s; VAD+ASR on ranks %s" % (roles["vision"], roles["audio"])
        wl = ("configs[4]: Mistral-7B fp16" if f16 else "configs[1] weights (Q4_K_M)") + " layer-sharded, 64-token prefill + 128-token greedy decode per cycle, %d row groups of %d" % (G, B)
        if combined:
            wl += ", fused with YOLOv8n 640x640 + VAD / Whisper-tiny.en (1 s PCM) on the spare GPUs"
        print(json.dumps({"metric": METRIC, "value": round(cycles * args.steps / elapsed, 3), "unit": "cycles/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / args.steps, 2),
                          **({} if COLL_CUDA else {"rehearsal": "TK_BENCH_SHARE_GPU=1: all ranks share ONE GPU; not a scaling measurement"}),
                          "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                          "dtype": "f16 weights x f16-rounded f32 activations, fp32 MFMA chain" if f16 else DTYPE, "data": "synthetic",
                          "config": {"workload": wl, "concurrent_cycles": cycles, "layers_per_stage": [bounds[r + 1] - bounds[r] for r in range(n_st)],
                                     "roles": roles, "parallelism": par},
                          **({"perception_handover": {"what": "detections (<= 20 per frame) + ASR token ids of every batch sent to the pipeline's stage 0 per step as fixed-size isend / irecv messages, required before its generate",
                                                      "bytes_per_step": ex.bytes_last, "checksum": ex.checksum}} if ex is not None and rank == roles["llm"][0] else {}),
                          "llm_tok_per_s": round(cycles * N * args.steps / elapsed, 1)}))
    if ex is not None:
        ex.finish()
    dist.destroy_process_group()


def model_per_gpu_roles(world):
    """rank 0 = the LLM; with 2 GPUs rank 1 runs detector + VAD + ASR; with more, the odd ranks run the detector and the
    even ranks (> 0) VAD + ASR.  Returns (vision ranks, audio ranks)."""
    per = list(range(1, world))
    if world <= 2:
        return per, per
    return [r for r in per if r % 2 == 1], [r for r in per if r % 2 == 0]


def run_model_per_gpu(args, tk, D, dist, hp, rank, local_rank, world, G, B, P, N):
    """SURVEY.md §8e rows "2" / "4" / "8": rank 0's GPU holds the LLM and decodes ALL cycles of the job; ranks 1.. run the detector
    (odd ranks) and VAD + ASR (even ranks, all of them when world == 2) for those cycles.  One step = G x B cycles through every
    stream once; the job's value = those cycles / max-over-ranks time.  What crosses GPUs in the reference's design is <= 20 detections
    and a text string per cycle (src/cortex/tk_cortex_main.c:1224-1237, 1323-1345): every step ends with that hand-over
    (D.PerceptionExchange: one gather to rank 0 inside the timed region) and rank 0 generates for a batch only once its results are there
    — the software pipeline of CycleBench (the LLM of batch k next to the perception of batch k + 1) with its dependency in place."""
    cycles = G * B
    vis, aud = model_per_gpu_roles(world)

    def barrier():
        D.barrier(dist, cuda=COLL_CUDA)

    ex = D.PerceptionExchange(dist, dst=0, plan=D.perception_plan(vis, aud, cycles, args.asr_steps), device="cuda" if COLL_CUDA else None)
    if rank == 0:
        model = tk.LlmModel(hp, device=local_rank).fill_synthetic(4)
        cb = CycleBench(tk, model, G, B, P, N, False, rank, local_rank, args.perception_batch, args.asr_steps)
        r = cb.run(args.steps, args.warmup, barrier, ex)
        role = "llm"
    else:
        share_v = -(-cycles // len(vis)) if rank in vis else 0
        share_a = -(-cycles // len(aud)) if rank in aud else 0
        cb = PerceptionBench(tk, share_v, share_a, rank, local_rank, args.perception_batch, args.asr_steps)
        r = cb.run(args.steps, args.warmup, barrier, ex)
        role = "perception"
    elapsed = D.max_over_ranks(dist, r["elapsed"], cuda=COLL_CUDA)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": round(cycles * args.steps / elapsed, 3), "unit": "cycles/s", "n_gpus": world, **({} if COLL_CUDA else {"rehearsal": "TK_BENCH_SHARE_GPU=1: all ranks share ONE GPU; not a scaling measurement"}), "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "strong",
                          "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
                          "config": {"workload": "configs[3]/[4] fused cycle, one model per GPU (SURVEY.md 8e)", "concurrent_cycles": cycles, "decode_groups": G,
                                     "rows_per_llm_pass": B, "prompt_tokens": P, "decode_tokens": N,
                                     "parallelism": "model-per-gpu x%d: LLM on rank 0; detector on ranks %s; VAD+ASR on ranks %s" % (world, vis, aud)},
                          "llm_rank_ms_per_step": round(1000.0 * r["elapsed"] / args.steps, 2), "role_of_rank0": role,
                          "perception_handover": {"what": "detections (<= 20 per frame) + ASR token ids of every batch sent to rank 0 per step as fixed-size isend / irecv messages, required before its generate",
                                                  "bytes_per_step": ex.bytes_last, "checksum": ex.checksum}}))
    cb.close()
    ex.finish()
    dist.destroy_process_group()


class PerceptionBench:
    """the detector and / or ASR + VAD streams alone, for a share of the job's cycles (model-per-gpu placement)"""

    def __init__(self, tk, n_frames, n_utts, rank, device, perception_batch, asr_steps):
        self.n_frames, self.n_utts, self.asr_steps = n_frames, n_utts, asr_steps
        self.PB = max(1, perception_batch)
        rng = np.random.default_rng(1 + rank)
        self.det = self.asr = self.vad = None
        if n_frames:
            self.det = tk.ObjectDetector(model="synthetic://yolov8n?seed=5&cls_bias=-0.45", width=640, height=640, conf=0.5, iou=0.5, device=device,
                                         max_batch=min(self.PB, n_frames))
            self.frames = [rng.integers(0, 256, (640, 640, 3), dtype=np.uint8) for _ in range(min(n_frames, 4 * self.PB))]
        if n_utts:
            self.asr = tk.Asr(hp=tk.WHISPER_TINY_EN(), seed=6, device=device, max_batch=min(self.PB, n_utts))
            self.vad = tk.Vad()
            self.pcm = np.clip(rng.normal(0, 3000, (min(n_utts, 4 * self.PB), 16000)), -32768, 32767).astype(np.int16)

    def _vision(self):
        self.dets = []
        for i in range(0, self.n_frames, self.PB):
            n = min(self.PB, self.n_frames - i)
            self.dets.extend(self.det.detect_batch([self.frames[(i + k) % len(self.frames)] for k in range(n)]))

    def _audio(self):
        for b in range(self.n_utts):
            self.vad.reset()
            self.vad.process_with_events(self.pcm[b % len(self.pcm)])
        toks = []
        for i in range(0, self.n_utts, self.PB):
            n = min(self.PB, self.n_utts - i)
            toks.append(self.asr.transcribe_tokens(self.pcm[[(i + k) % len(self.pcm) for k in range(n)]], self.asr_steps, want_aux=False)[0])
        self.toks = np.concatenate(toks) if toks else None

    def step(self):
        """one batch through this rank's streams; returns what the LLM's first rank needs of it (D.pack_perception)"""
        th = []
        self.dets, self.toks = None, None
        if self.det:
            th.append(threading.Thread(target=self._vision))
        if self.asr:
            th.append(threading.Thread(target=self._audio))
        for t in th:
            t.start()
        for t in th:
            t.join()
        from trackiellm_amd import dist as D
        return D.pack_perception(self.dets, self.toks)

    def run(self, steps, warmup, barrier, exchange=None):
        for _ in range(warmup):
            res = self.step()
            if exchange is not None:
                exchange.hand_over(res)
        barrier()
        t0 = time.time()
        for _ in range(steps):
            res = self.step()
            if exchange is not None:
                exchange.hand_over(res)  # the step's results travel to the LLM's first rank inside the timed region
        barrier()
        return {"elapsed": time.time() - t0}

    def close(self):
        for o in (self.det, self.asr, self.vad):
            if o is not None:
                o.close()


if __name__ == "__main__":
    main()
