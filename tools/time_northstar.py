#!/usr/bin/env python3
"""The north-star point alone (developer tool, needs an MI355X): G decode groups x B rows per pass, the fused workload of bench.py's
`north_star_point` (or --llm-only), timed over a few steps; prints cycles/s, the decode window's and the whole run's HBM fraction by
SURVEY.md 8(d)'s wall formula.  Environment switches of the library apply (they are read when a pass is recorded).
    python tools/time_northstar.py [--groups 3] [--rows 16] [--steps 3] [--llm-only]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import trackiellm_amd as tk  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--groups", type=int, nargs="+", default=[3])
ap.add_argument("--rows", type=int, default=16)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--llm-only", action="store_true")
ap.add_argument("--tag", default="")
a = ap.parse_args()
P, N = 64, 128
model = tk.LlmModel(tk.MISTRAL_7B(), device=0).fill_synthetic(4)
for G in a.groups:
    cb = bench.CycleBench(tk, model, G, a.rows, P, N, not a.llm_only, 0, 0, 64, 16)
    r = cb.run(a.steps, 1)
    cycles = G * a.rows
    n_pre = G * (-(-(a.rows * (P - 1)) // 256) + 1)
    passes = G * N + n_pre
    whole = model.weight_bytes * passes * a.steps / r["elapsed"] / 8e12
    window = (model.weight_bytes + a.rows * 131072 * (P + N / 2.0)) * (G * 1000.0 / r["decode_ms_per_step"]) / 8e12
    print(f"{a.tag} groups {G} x rows {a.rows} {'llm-only' if a.llm_only else 'fused'}: {cycles * a.steps / r['elapsed']:.2f} cycles/s, decode {r['decode_ms_per_step']:.3f} ms per step and group, "
          f"decode window {window:.4f}, whole run {whole:.4f} of 8 TB/s (prefill {r['prefill_s'] * 1e3:.1f} ms, decode {r['decode_s'] * 1e3:.1f} ms per step)", flush=True)
    cb.close()


def free_running(G, rows, steps, fused):
    """the same work with the groups NOT joined per step: every group thread runs `steps` cycles batches (perception of its rows beside its prefill +
    decode) on its own clock, group g starting g / G of a step late — prefill passes (compute-bound, 256 rows) of one group then fall beside the decode
    passes (HBM-bound, 16 rows) of the others instead of all groups prefilling, then all decoding, together"""
    import threading
    import time
    import numpy as np
    hp = model.hparams
    sess = [tk.LlmSession(model, rows, P + N + 8) for _ in range(G)]
    prompts = []
    for g in range(G):
        pr = np.stack([bench.splitmix_tokens(3 + 1000 * (g * rows + s), P, 3, hp.vocab) for s in range(rows)])
        pr[:, 0] = 1
        prompts.append(pr)
    det = asr = vad = None
    if fused:
        det = [tk.ObjectDetector(model="synthetic://yolov8n?seed=5&cls_bias=-0.45", width=640, height=640, conf=0.5, iou=0.5, max_batch=rows) for _ in range(G)]
        asr = [tk.Asr(hp=tk.WHISPER_TINY_EN(), seed=6, max_batch=rows) for _ in range(G)]
        vad = [tk.Vad() for _ in range(G)]
        rng = np.random.default_rng(1)
        frames = [rng.integers(0, 256, (640, 640, 3), dtype=np.uint8) for _ in range(rows)]
        pcm = np.clip(np.random.default_rng(2).normal(0, 3000, (rows, 16000)), -32768, 32767).astype(np.int16)

    def perception(g):
        det[g].detect_batch(frames)
        for b in range(rows):
            vad[g].reset()
            vad[g].process_with_events(pcm[b])
        asr[g].transcribe_tokens(pcm, 16, want_aux=False)

    def cycle_batch(g):
        th = threading.Thread(target=perception, args=(g,)) if fused else None
        if th:
            th.start()
        sess[g].prefill(prompts[g])
        sess[g].decode(rows, N)
        if th:
            th.join()

    for g in range(G):  # warm-up: captures, engines
        cycle_batch(g)
    t_one = time.time()
    cycle_batch(0)
    t_one = time.time() - t_one
    done = [0.0] * G

    def run(g):
        time.sleep(g * 1.6 * t_one / G)  # a lone batch takes t_one; G concurrent ones ~1.6 x that: spread the starts over one such step
        for _ in range(steps):
            cycle_batch(g)
        done[g] = time.time()

    th = [threading.Thread(target=run, args=(g,)) for g in range(G)]
    t0 = time.time()
    for t in th:
        t.start()
    for t in th:
        t.join()
    elapsed = max(done) - t0
    n_pre = G * (-(-(rows * (P - 1)) // 256) + 1)
    passes = G * N + n_pre
    whole = model.weight_bytes * passes * steps / elapsed / 8e12
    print(f"{a.tag} FREE-RUNNING groups {G} x rows {rows} {'fused' if fused else 'llm-only'}: {G * rows * steps / elapsed:.2f} cycles/s, whole run {whole:.4f} of 8 TB/s "
          f"({steps} cycle batches per group, staggered starts included in the time)", flush=True)
    for s in sess:
        s.close()


if os.environ.get("TK_NS_FREE") == "1":
    for G in a.groups:
        free_running(G, a.rows, max(a.steps, 4), not a.llm_only)
