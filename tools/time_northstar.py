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
