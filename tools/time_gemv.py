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
    ap.add_argument("--real-activations", action="store_true", help="run one real pass first: the activation images then hold real data (the "
                    "chip's clock under an MFMA loop depends on the operand values)")
    ap.add_argument("--only", default="", help="comma-separated shape names")
    ap.add_argument("--stamps", action="store_true", help="TK_MI355X_LIB is a -DTK_G32_CLOCK=1 build: print the in-kernel clock per shape")
    ap.add_argument("--zero-weights", action="store_true", help="diagnostic: all-zero gate / up / o / q / k blocks (same instruction stream, least MFMA energy)")
    a = ap.parse_args()
    model = tk.LlmModel(tk.MISTRAL_7B(), device=0).fill_synthetic(4)
    n_layer = model.hparams.n_layer
    if a.zero_weights:
        import numpy as np
        hp = model.hparams
        qd, kv = hp.n_head * hp.head_dim, hp.n_kv_head * hp.head_dim
        for l in range(n_layer):
            for which, rows, cols in ((1, qd, hp.d_model), (2, kv, hp.d_model), (4, hp.d_model, qd), (6, hp.d_ff, hp.d_model), (7, hp.d_ff, hp.d_model)):
                model.set_tensor(l, which, 12, np.zeros(rows * cols // 256 * 144, np.uint8))
    sess = tk.LlmSession(model, a.rows, 64)
    if a.real_activations:
        import numpy as np
        rng = np.random.default_rng(1)
        sess.forward(np.arange(a.rows), np.zeros(a.rows, np.int32), rng.integers(0, model.hparams.vocab, a.rows), want_logits=False)
    q6 = [l for l in range(n_layer) if l < n_layer // 8 or l >= 7 * n_layer // 8 or (l - n_layer // 8) % 3 == 2]
    q4 = [l for l in range(n_layer) if l not in q6]
    shapes = [("gate_up", 0, 0, n_layer), ("down_q6", 1, q6[0], len(q6)), ("down_q4", 1, q4[0], len(q4)), ("qkv_q6", 2, q6[0], len(q6)),
              ("qkv_q4", 2, q4[0], len(q4)), ("o", 4, 0, n_layer), ("lm_head", 3, 0, 1)]
    tot_ms = tot_b = 0.0
    for name, which, layer, n in shapes:
        if a.only and name not in a.only.split(","):
            continue
        ms, b = sess.time_gemv(layer, which, a.rows, a.iters)
        tot_ms += ms * n
        tot_b += b * n
        print(f"{name:8s} {ms * 1e3:8.2f} us  {b / ms / 1e6:8.1f} GB/s  x{n}")
        if a.stamps:  # a TK_G32_CLOCK diagnostic build: the last launch's per-workgroup (cycles, 100 MHz ticks, blocks) of the K loop
            import ctypes
            import numpy as np
            L = ctypes.CDLL(os.environ["TK_MI355X_LIB"])
            st = np.zeros((1024, 8), np.uint64)
            assert L.tk_debug_g32_stamps(st.ctypes.data_as(ctypes.c_void_p), 1024) == 0
            st = st[: int(st[0, 3])] if 0 < st[0, 3] <= 1024 else st[:0]
            if len(st):
                cyc, real, nb = np.median(st[:, 0].astype(float)), np.median(st[:, 1].astype(float)), float(st[0, 2])
                if hasattr(L, "tk_debug_g32_seg"):
                    sg = np.zeros((1024, 8, 4), np.uint64)
                    assert L.tk_debug_g32_seg(sg.ctypes.data_as(ctypes.c_void_p), 1024) == 0
                    sg = sg[: len(st)].astype(float) / nb
                    if sg.sum() > 0:
                        m = np.zeros((8, 4))  # [wave slot: pair + 4 * row half][segment], cycles per block; a workgroup fills the four slots of its half
                        for w in range(8):
                            rows = sg[:, w][sg[:, w].sum(axis=1) > 0]
                            if len(rows):
                                m[w] = np.median(rows, axis=0)
                        for w in range(8):
                            print(f"         wave {w}: unpack {m[w, 0]:6.0f}  tiles {m[w, 1]:6.0f}  s_waitcnt(0) {m[w, 2]:6.0f}  barrier {m[w, 3]:6.0f}  = {m[w].sum():6.0f} cycles per block")
                ab = st[:, 4:8].astype(float)
                t0 = ab[:, 0].min()
                us = (ab - t0) / 100.0
                print(f"         timeline of the last launch (us after the first workgroup's entry): entry median {np.median(us[:, 0]):.1f} max {us[:, 0].max():.1f}; loop start median "
                      f"{np.median(us[:, 1]):.1f}; loop end median {np.median(us[:, 2]):.1f} p90 {np.percentile(us[:, 2], 90):.1f} max {us[:, 2].max():.1f}; exit median {np.median(us[:, 3]):.1f} "
                      f"max {us[:, 3].max():.1f}; K loop us min {(us[:, 2] - us[:, 1]).min():.1f} median {np.median(us[:, 2] - us[:, 1]):.1f} max {(us[:, 2] - us[:, 1]).max():.1f}")
                print(f"         K loop per workgroup (median of {len(st)}): {cyc:.0f} cycles, {real / 100:.2f} us -> in-kernel clock {cyc / real * 100:.0f} MHz, "
                      f"{cyc / nb:.0f} cycles per 256-k block ({nb:.0f} blocks)")
    print(f"step     {tot_ms * 1e3:8.1f} us  {tot_b / tot_ms / 1e6:8.1f} GB/s  frac {tot_b / tot_ms / 1e6 / 8000:.4f}", flush=True)


if __name__ == "__main__":
    main()
