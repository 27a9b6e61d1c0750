#!/usr/bin/env python3
"""Decode-step time of one session at a given pass width (developer tool, needs an MI355X): a 64-token prompt per sequence, then
N greedy steps through the captured pass — ms per step (HIP events around the loop) and the weight-stream rate it corresponds to.
    python tools/time_decode.py [rows] [steps]
Environment switches read when a pass is recorded: TK_MI355X_NO_TAIL=1, TK_MI355X_NO_FUSE=1, TK_MI355X_NO_GRAPH=1."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trackiellm_amd as tk  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 128
model = tk.LlmModel(tk.MISTRAL_7B(), device=0).fill_synthetic(4)
wb = tk.lib().tk_mi355x_llm_model_weight_bytes
wb.restype = __import__("ctypes").c_uint64
nbytes = wb(model.h)
sess = tk.LlmSession(model, rows, 64 + 3 * steps + 72)
rng = np.random.default_rng(1)
prompts = rng.integers(3, model.hparams.vocab, (rows, 64)).astype(np.int32)
prompts[:, 0] = 1
sess.prefill(prompts)
sess.decode(rows, 8)
for rep in range(3):
    toks, ms = sess.decode(rows, steps)
    print(f"{rows} rows: {ms:.3f} ms per decode step, {rows / ms * 1e3:.0f} tok/s, weights at {nbytes / ms / 1e9:.2f} TB/s = {nbytes / ms / 1e9 / 8:.3f} of 8 TB/s "
          f"(NO_TAIL={os.environ.get('TK_MI355X_NO_TAIL', '0')} NO_FUSE={os.environ.get('TK_MI355X_NO_FUSE', '0')})", flush=True)
