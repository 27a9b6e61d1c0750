#!/usr/bin/env python3
"""bench.py — headline benchmark of the fused perception->reasoning hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input: B concurrent cortex
cycles, each = (640x640 frame detect + 1 s PCM VAD/ASR, when those streams are enabled) +
64-token prompt prefill + 128-token greedy Mistral-7B Q4_K_M decode.  B is stated in
config (the weight stream is shared by the cycles decoded together).

    python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches one rank per GPU with torch.distributed.run; ranks are
independent replicas of the cycle batch (no data-path collective, weak scaling).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
INT8_PEAK_TOPS = 5000.0  # dense int8 MFMA, 2x the bf16 rate (same guide, matrix cores)


def splitmix_tokens(seed, n, lo, hi):
    rng = np.random.default_rng(seed)
    return rng.integers(lo, hi, n).astype(np.int32)


def cpu_baseline(hp, max_tokens=3):
    """oracle ('port') on the host cores: bounded sample of the same Mistral-7B workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, int(os.environ.get("TK_BENCH_CPU_CORES", "16")))  # the GPU box grants a 16-core share per GPU
    os.environ["OMP_NUM_THREADS"] = str(cores)
    cfg = O.LlmConfig(n_layer=hp.n_layer, d_model=hp.d_model, n_head=hp.n_head, n_kv_head=hp.n_kv_head, head_dim=hp.head_dim,
                      d_ff=hp.d_ff, vocab=hp.vocab, max_ctx=16, max_seq=1, rms_eps=hp.rms_eps, rope_theta=hp.rope_theta,
                      ks_qkv=hp.ks_qkv, ks_o=hp.ks_o, ks_gateup=hp.ks_gateup, ks_down=hp.ks_down, ks_out=hp.ks_out)
    t0 = time.time()
    orc = O.OracleLlm(cfg, seed=4)
    t_synth = time.time() - t0
    toks = []
    cur = 1
    t0 = time.time()
    for i in range(max_tokens):
        _, am = orc.forward([0], [i], [cur], want_logits=False)
        cur = int(am[0])
        toks.append(cur)
    dt = (time.time() - t0) / max_tokens
    orc.close()
    return dt, t_synth, toks, cores


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
    ap.add_argument("--sessions", type=int, default=3,
                    help="independent decode groups of --batch cycles run concurrently on their own HIP streams (fills the "
                         "launch/ramp bubbles of one group with another group's kernels); concurrent cycles = sessions * batch")
    ap.add_argument("--llm-only", action="store_true", help="configs[1] only: leave the detector / ASR / VAD streams out (marked in config)")
    ap.add_argument("--pipeline", action="store_true",
                    help="opt-in: the ranks form ONE layer-sharded LLM pipeline (RCCL send / recv of the residual stream between consecutive "
                         "GPUs, SURVEY.md 8e) instead of independent replicas; LLM stream only; --sessions row groups keep the stages busy")
    ap.add_argument("--roofline-only", action="store_true",
                    help="only the isolated per-shape timing of the dominant kernel (the roofline object); profile THIS command with "
                         "rocprofv3 --kernel-trace to compare its kernel durations with the HIP-event numbers (tools/roofline_check.py)")
    ap.add_argument("--perception-batch", type=int, default=64, help="frames / utterances per detector / ASR call")
    ap.add_argument("--asr-steps", type=int, default=16, help="forced greedy decoder steps per utterance (SURVEY.md 8d)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    from trackiellm_amd import dist as D
    if world > 1:
        dist = D.init("nccl", local_rank)

    import trackiellm_amd as tk
    if tk.lib().tk_mi355x_device_count() <= local_rank:
        raise SystemExit("bench.py needs one MI355X per rank: no fallback path exists")
    tk.lib().tk_mi355x_set_default_device(local_rank)

    B, P, N = args.batch, args.prompt, args.decode
    hp = tk.MISTRAL_7B()
    hp.n_layer = args.layers
    t0 = time.time()
    model = tk.LlmModel(hp, device=local_rank).fill_synthetic(4)
    hp = model.hparams
    G = max(1, args.sessions)
    sessions = [tk.LlmSession(model, B, P + N + 8) for _ in range(G)]
    sess = sessions[0]
    t_load = time.time() - t0
    prompts = []
    for g in range(G):
        pr = np.stack([splitmix_tokens(3 + 1000 * ((rank * G + g) * B + s), P, 3, hp.vocab) for s in range(B)])
        pr[:, 0] = 1  # BOS
        prompts.append(pr)
    import threading

    if args.pipeline:
        # one model, layers split over the ranks; every rank builds the same weights and walks the same pass order
        if dist is None:
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
            dist = D.init("gloo")
        cuda_t = world > 1
        big = tk.LlmSession(model, G * B, P + N + 8)
        pipe = D.LlmPipeline(dist, big, hp.n_layer, hp.d_model, cuda_tensors=cuda_t)
        same = [np.stack([splitmix_tokens(3 + 1000 * (g * B + s_), P, 3, hp.vocab) for s_ in range(B)]) for g in range(G)]
        for pr in same:
            pr[:, 0] = 1
        for _ in range(args.warmup):
            pipe.generate(same, N)
        D.barrier(dist, cuda=cuda_t)
        t0 = time.time()
        for _ in range(args.steps):
            pipe.generate(same, N)
        D.barrier(dist, cuda=cuda_t)
        elapsed = D.max_over_ranks(dist, time.time() - t0, cuda=cuda_t)
        if rank == 0:
            print(json.dumps({"metric": "cortex cycles/sec (frame+1s audio+128 tok)", "value": round(G * B * args.steps / elapsed, 3), "unit": "cycles/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / args.steps, 2),
                              "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int8 x q4_K/q6_K (i32 acc, f32 scales)",
                              "data": "synthetic",
                              "config": {"workload": "configs[1] LLM stream only, layer-sharded: Mistral-7B Q4_K_M, 64-token prefill + 128-token greedy decode "
                                                     "per cycle, host-driven passes (no hipGraph), %d row groups of %d" % (G, B),
                                         "concurrent_cycles": G * B, "layers_per_rank": [pipe.bounds[r + 1] - pipe.bounds[r] for r in range(world)],
                                         "parallelism": "pipeline x%d (RCCL send/recv of [rows, 4096] fp32 between consecutive stages)" % world},
                              "llm_tok_per_s": round(G * B * N * args.steps / elapsed, 1)}))
        dist.destroy_process_group()
        return

    # perception streams: one 640x640 frame and 1 s of PCM per cycle, their own HIP streams, driven from host threads
    if args.roofline_only:
        args.llm_only, args.steps, args.warmup, args.no_cpu_baseline = True, 0, 0, True
    fused = not args.llm_only
    perc_ms = {"vision": [], "audio": []}
    if fused:
        PB = min(B, args.perception_batch)  # frames / utterances per detector / ASR call
        det = tk.ObjectDetector(model="synthetic://yolov8n?seed=5&cls_bias=-0.45", width=640, height=640, conf=0.5, iou=0.5,
                                device=local_rank, max_batch=PB)
        asr = tk.Asr(hp=tk.WHISPER_TINY_EN(), seed=6, device=local_rank, max_batch=PB)
        vad = tk.Vad()
        frng = np.random.default_rng(1 + rank)
        frames = [frng.integers(0, 256, (640, 640, 3), dtype=np.uint8) for _ in range(G * B)]
        prng = np.random.default_rng(2 + rank)
        pcm = np.clip(prng.normal(0, 3000, (G * B, 16000)), -32768, 32767).astype(np.int16)
        n_dets = [0]

        def vision_pass():  # one frame per concurrent cycle, PB frames per detector call
            t = time.time()
            n = 0
            for i in range(0, G * B, PB):
                n += sum(len(r) for r in det.detect_batch(frames[i:i + PB]))
            n_dets[0] = n
            perc_ms["vision"].append(1000 * (time.time() - t))

        def audio_pass():  # one second of PCM per concurrent cycle
            t = time.time()
            for b in range(G * B):
                vad.reset()
                vad.process_with_events(pcm[b])
            for i in range(0, G * B, PB):
                asr.transcribe_tokens(pcm[i:i + PB], args.asr_steps, want_aux=False)
            perc_ms["audio"].append(1000 * (time.time() - t))

        def perception_async():
            th = [threading.Thread(target=vision_pass), threading.Thread(target=audio_pass)]
            for t in th:
                t.start()
            return th

    def barrier():
        if dist is not None:
            D.barrier(dist, cuda=True)

    def one_step():
        # software pipeline: the LLM consumes the perception results of THIS cycle batch (produced during the previous
        # step) while the detector / ASR / VAD streams already work on the next batch; every step runs all streams once
        th = perception_async() if fused else []
        res = [None] * G

        def llm_group(g):
            t_a = time.time()
            sessions[g].prefill(prompts[g])
            t_b = time.time()
            toks, ms_step = sessions[g].decode(B, N)
            res[g] = (toks, t_b - t_a, time.time() - t_b, ms_step)

        lt = [threading.Thread(target=llm_group, args=(g,)) for g in range(1, G)]
        for t in lt:
            t.start()
        llm_group(0)
        for t in lt + th:
            t.join()
        return res[0][0], float(np.mean([r[1] for r in res])), float(np.mean([r[2] for r in res])), float(np.mean([r[3] for r in res]))

    if fused:
        for t in perception_async():  # primes the pipeline (perception of the first timed batch)
            t.join()
        perc_ms["vision"].clear(); perc_ms["audio"].clear()
    for _ in range(args.warmup):
        one_step()
    perc_ms["vision"].clear(); perc_ms["audio"].clear()
    barrier()
    t0 = time.time()
    pre_s = dec_s = 0.0
    ms_steps = []
    for _ in range(args.steps):
        toks, a, b, ms = one_step()
        pre_s += a
        dec_s += b
        ms_steps.append(ms)
    barrier()
    elapsed = time.time() - t0
    if dist is not None:
        elapsed = D.max_over_ranks(dist, elapsed, cuda=True)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (k_gemv_w4a8), HIP events on the session stream, live ----
    n_layer = hp.n_layer
    q6_layers = [l for l in range(n_layer) if l < n_layer // 8 or l >= 7 * n_layer // 8 or (l - n_layer // 8) % 3 == 2]
    q4_layers = [l for l in range(n_layer) if l not in q6_layers]

    def gemv_roofline(rows):
        shapes = {}
        total_ms = total_bytes = 0.0
        launches = 0
        for name, which, per_layer in (("gate_up", 0, True), ("down", 1, True), ("qkv", 2, True), ("o", 4, True), ("lm_head", 3, False)):
            groups = ([("q6", q6_layers), ("q4", q4_layers)] if which in (1, 2) else [("", list(range(n_layer)))]) if per_layer else [("", [0])]
            for tag, layers in groups:
                if not layers:
                    continue
                ms, nbytes = sess.time_gemv(layers[0], which, rows, 50)
                cnt = len(layers) if per_layer else 1
                shapes[name + ("_" + tag if tag else "")] = {"ms": round(ms, 5), "GBps": round(nbytes / ms / 1e6, 1), "launches_per_step": cnt}
                total_ms += ms * cnt
                total_bytes += nbytes * cnt
                launches += cnt
        achieved = total_bytes / total_ms / 1e6  # GB/s
        kernel = "k_gemm_w4a8" if rows > 32 else "k_gemv_w4a8"  # > 32 rows: the K-streamed batched variant of the same arithmetic
        return {"bound": "hbm", "kernel": kernel, "rows_per_pass": rows, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes_per_launch": round(total_bytes / launches),
                "avg_launch_ms": round(total_ms / launches, 5), "launches_per_decode_step": launches, "per_shape": shapes}

    roofline = gemv_roofline(B)
    # HBM traffic per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE collected separately, gfx950 correction applied)
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_gemm_b256.json" if B > 128 else "r01_pmc_gemm_b128.json" if B > 32 else "r01_pmc_gemv.json")
    if os.path.exists(pmc):
        pj = json.load(open(pmc))
        if pj.get("rows_per_pass") == B:
            roofline["traffic"] = pj.get("hbm_bytes_per_average_launch")
    # the same launches priced against the matrix cores: 2 int8 ops per (row, weight); 16x16x64 i8 MFMA = 2x the bf16 rate
    qd, kvd = hp.n_head * hp.head_dim, hp.n_kv_head * hp.head_dim
    w_step = hp.n_layer * (hp.d_model * (qd + 2 * kvd) + qd * hp.d_model + 3 * hp.d_model * hp.d_ff) + hp.vocab * hp.d_model
    tops = 2.0 * B * w_step / (roofline["avg_launch_ms"] * 1e-3 * roofline["launches_per_decode_step"]) / 1e12
    # which roof bounds the launch set: algorithmic int8 ops per algorithmic byte against the ridge (5000 TOP/s / 8 TB/s = 625 op/B)
    intensity = 2.0 * B * w_step / (roofline["algorithmic_bytes_per_launch"] * roofline["launches_per_decode_step"])
    roofline["int8_ops_per_byte"] = round(intensity, 1)
    if intensity > INT8_PEAK_TOPS * 1e12 / (HBM_PEAK_GBS * 1e9):
        roofline["hbm_view"] = {k: roofline[k] for k in ("achieved", "peak", "unit", "frac")}
        roofline.update({"bound": "mfma", "achieved": round(tops, 1), "peak": INT8_PEAK_TOPS, "unit": "TOP/s", "frac": round(tops / INT8_PEAK_TOPS, 4)})
    else:
        roofline["int8_tops"] = round(tops, 1)
        roofline["int8_peak_tops"] = INT8_PEAK_TOPS
    # the same kernel at 16 rows per pass (one MFMA M-tile): less integer work per weight byte, closer to the HBM bound
    roofline_16 = gemv_roofline(16) if B > 16 else None

    if args.roofline_only:
        print(json.dumps({"metric": "cortex cycles/sec (frame+1s audio+128 tok)", "value": None, "unit": "cycles/s", "n_gpus": world,
                          "note": "roofline-only run: no timed steps", "roofline": roofline,
                          "roofline_16_rows": roofline_16 and {k: roofline_16[k] for k in ("rows_per_pass", "achieved", "frac", "avg_launch_ms", "per_shape")}}))
        return
    value = D.aggregate_throughput(G * B, args.steps, world, elapsed)
    dec_ms = float(np.mean(ms_steps))
    out = {
        "metric": "cortex cycles/sec (frame+1s audio+128 tok)", "value": round(value, 3), "unit": "cycles/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / args.steps, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int8 x q4_K/q6_K (i32 acc, f32 scales)",
        "data": "synthetic",
        "config": {"workload": ("configs[3] fused cycle: YOLOv8n 640x640 frame (preprocess+network+NMS) + 1 s PCM (VAD + Whisper-tiny.en "
                                "log-mel/encoder/%d forced decoder steps) + Mistral-7B Q4_K_M 64-token prefill and 128-token greedy decode, "
                                "3 concurrent HIP streams" % args.asr_steps) if fused else
                               "configs[1]: Mistral-7B Q4_K_M, 64-token prefill + 128-token greedy decode per cycle (LLM stream only)",
                   "concurrent_cycles_per_gpu": G * B, "decode_groups": G, "rows_per_llm_pass": B, "prompt_tokens": P, "decode_tokens": N, "layers": hp.n_layer,
                   "k_split": [hp.ks_qkv, hp.ks_o, hp.ks_gateup, hp.ks_down], "parallelism": f"replicas x{world}"},
        "llm_tok_per_s": round(G * B * world * N / (dec_s / args.steps), 1),
        "decode_ms_per_step": round(dec_ms, 4), "prefill_s_per_cycle_batch": round(pre_s / args.steps, 4),
        "model_load_s": round(t_load, 2), "weight_bytes_per_decode_step": int(model.weight_bytes),
        "roofline": roofline,
        # SURVEY.md 8d's whole-step view: (weights + B * 128 KiB * mean context of KV per row) per decode step over 8 TB/s, all groups in flight
        "llm_decode_step_roofline": {"bytes_per_step": int(model.weight_bytes + B * 131072 * (P + N / 2.0)),
                                     "steps_per_s": round(G * 1000.0 / dec_ms, 1),
                                     "frac": round((model.weight_bytes + B * 131072 * (P + N / 2.0)) * (G * 1000.0 / dec_ms) / (HBM_PEAK_GBS * 1e9), 4)},
    }
    if roofline_16 is not None:
        out["roofline_16_rows"] = {k: roofline_16[k] for k in ("rows_per_pass", "achieved", "peak", "unit", "frac", "avg_launch_ms", "per_shape")}
    if fused:
        out["perception"] = {"vision_ms_per_batch": round(float(np.mean(perc_ms["vision"])), 2), "audio_ms_per_batch": round(float(np.mean(perc_ms["audio"])), 2),
                             "detections_last_batch": n_dets[0], "overlapped_with_llm": True, "dtype": "f32 (exact fp32 MFMA chain)"}
    if args.layers != 32:
        out["invalid"] = "debug run with fewer layers"
    if not args.no_cpu_baseline and args.layers == 32 and world == 1:  # the CPU baseline is a rank-0, N = 1 leg
        s_per_tok, t_synth, otoks, cores = cpu_baseline(hp)
        # parity spot-check on the same weights: 1 sequence, BOS then greedy
        chk = tk.LlmSession(model, 1, 16)
        _, am = chk.forward([0], [0], [1], want_logits=False)
        gtoks = [int(am[0])]
        for i in range(1, len(otoks)):
            _, am = chk.forward([0], [i], [gtoks[-1]], want_logits=False)
            gtoks.append(int(am[0]))
        out["cpu_baseline"] = {"value": round(1.0 / (s_per_tok * (P + N)), 5), "unit": "cycles/s", "cores": cores, "kind": "port",
                               "sample": f"oracle (CPU restatement, not llama.cpp), full 32-layer Mistral-7B Q4_K_M, batch 1, "
                                         f"{len(otoks)} tokens timed ({s_per_tok:.2f} s/token, weights synthesised in {t_synth:.0f} s); "
                                         f"cycle = {P}+{N} tokens extrapolated",
                               "tok_per_s": round(1.0 / s_per_tok, 3), "token_ids_match_gpu": gtoks == otoks}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
