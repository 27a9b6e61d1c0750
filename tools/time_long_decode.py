#!/usr/bin/env python3
"""Decode over a long context at few rows: `rows` sequences with a `ctx`-token prompt each, then 64 decode steps through the device-side loop —
developer tool, needs an MI355X.      python tools/time_long_decode.py 1:2000,1:3900,4:2000 [steps]
TK_MI355X_NO_LONG_ATT=1: the fused attention kernels whatever the context."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import trackiellm_amd as tk  # noqa: E402

cases = [tuple(int(v) for v in c.split(":")) for c in (sys.argv[1] if len(sys.argv) > 1 else "1:2000").split(",")]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
model = tk.LlmModel(tk.MISTRAL_7B()).fill_synthetic(4)
hp = model.hparams
rng = np.random.default_rng(9)
form = "fused kernels" if os.environ.get("TK_MI355X_NO_LONG_ATT") == "1" else "long-context form"
for rows, ctx in cases:
    sess = tk.LlmSession(model, rows, ctx + 2 * steps + 8)
    toks = rng.integers(3, hp.vocab, (rows, ctx)).astype(np.int32)
    first = sess.prefill(toks)
    ids0, _ = sess.decode(rows, steps)      # captures the passes
    ids1, ms = sess.decode(rows, steps)     # timed: positions ctx + steps .. ctx + 2 steps
    print("%d rows over %d .. %d positions: %.3f ms per decode step, %.0f tok/s  [%s]  ids %s" %
          (rows, ctx + steps, ctx + 2 * steps, ms, rows * 1000.0 / ms, form, ids1[:3, 0].tolist()), flush=True)
    sess.close()
model.close()
