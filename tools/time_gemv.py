#!/usr/bin/env python3
"""Per-shape timing of the W4A8 GEMV kernel on the Mistral-7B Q4_K_M plan (developer tool, needs an MI355X).

    python tools/time_gemv.py [--rows 32] [--iters 50]

One line per GEMV shape of a decode step: average launch time (HIP events around `iters` back-to-back launches that
cycle through the layers, so every launch streams cold HBM lines) and algorithmic GB/s.  A diagnostic build
(tools/build_ablate.sh) is selected with TK_MI355X_LIB=<path to .so>.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trackiellm_amd as tk  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=32)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    model = tk.LlmModel(tk.MISTRAL_7B(), device=0).fill_synthetic(4)
    n_layer = model.hparams.n_layer
    sess = tk.LlmSession(model, a.rows, 64)
    q6 = [l for l in range(n_layer) if l < n_layer // 8 or l >= 7 * n_layer // 8 or (l - n_layer // 8) % 3 == 2]
    q4 = [l for l in range(n_layer) if l not in q6]
    shapes = [("gate_up", 0, 0, n_layer), ("down_q6", 1, q6[0], len(q6)), ("down_q4", 1, q4[0], len(q4)), ("qkv_q6", 2, q6[0], len(q6)),
              ("qkv_q4", 2, q4[0], len(q4)), ("o", 4, 0, n_layer), ("lm_head", 3, 0, 1)]
    tot_ms = tot_b = 0.0
    for name, which, layer, n in shapes:
        ms, b = sess.time_gemv(layer, which, a.rows, a.iters)
        tot_ms += ms * n
        tot_b += b * n
        print(f"{name:8s} {ms * 1e3:8.2f} us  {b / ms / 1e6:8.1f} GB/s  x{n}")
    print(f"step     {tot_ms * 1e3:8.1f} us  {tot_b / tot_ms / 1e6:8.1f} GB/s  frac {tot_b / tot_ms / 1e6 / 8000:.4f}", flush=True)


if __name__ == "__main__":
    main()
