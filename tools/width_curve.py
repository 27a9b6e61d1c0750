#!/usr/bin/env python3
"""The W4A8 launch set of one decode step by rows per pass (developer tool, needs an MI355X): per width the Σ of the 129 launches' average
durations (tk_mi355x_llm_time_gemv: graph path, launches cycling through the layers = cold weights) and ONE byte count — the WEIGHT bytes
of SURVEY.md 8(d), 4.293 GB per step, the same definition as bench.py's `roofline.hbm_view` — over 8 TB/s.
    python tools/width_curve.py [rows ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import trackiellm_amd as tk  # noqa: E402

widths = [int(x) for x in sys.argv[1:]] or [16, 32, 33, 64, 65, 96, 97, 128, 129, 160, 161, 192, 193, 224, 225, 256]
model = tk.LlmModel(tk.MISTRAL_7B(), device=0).fill_synthetic(4)
hp = model.hparams
print("rows  kernel           step_ms  weight_TB/s  frac_of_8TB/s  gate_up_us  down_q4_us  qkv_q4_us  o_us  lm_head_us   (bytes = weights only, %.3f GB per step)" % (model.weight_bytes / 1e9))
for rows in widths:
    sess = tk.LlmSession(model, rows, 64)
    r = bench.gemv_roofline(sess, hp, rows, model.weight_bytes, iters=40)
    ps = r["per_shape"]
    step_ms = r["avg_launch_ms"] * r["launches_per_decode_step"]
    hv = r["hbm_view"]
    print("%4d  %-15s %8.3f  %11.3f  %13.4f  %10.2f  %10.2f  %9.2f  %4.2f  %10.2f" % (rows, r["kernel"], step_ms, hv["achieved"] / 1e3, hv["frac"], 1e3 * ps["gate_up"]["ms"],
          1e3 * ps["down_q4"]["ms"], 1e3 * ps["qkv_q4"]["ms"], 1e3 * ps["o"]["ms"], 1e3 * ps["lm_head"]["ms"]), flush=True)
    sess.close()
