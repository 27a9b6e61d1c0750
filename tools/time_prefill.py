#!/usr/bin/env python3
"""Prompt processing of the full Mistral-7B geometry: `nseq` prompts of `n_prompt` tokens through TkLlmSession::prefill (passes of up to 256
rows, several positions of a sequence per pass) — developer tool, needs an MI355X.
    python tools/time_prefill.py 1x448,1x1024,1x2048,4x64 [reps]      TK_MI355X_NO_PREFILL_ATT=1 for the one-workgroup-per-row attention"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import trackiellm_amd as tk  # noqa: E402

cases = [tuple(int(v) for v in c.split("x")) for c in (sys.argv[1] if len(sys.argv) > 1 else "1x448,1x1024").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
model = tk.LlmModel(tk.MISTRAL_7B()).fill_synthetic(4)
hp = model.hparams
rng = np.random.default_rng(5)
form = "k_attention (per row)" if os.environ.get("TK_MI355X_NO_PREFILL_ATT") == "1" else "k_attention_prefill"
for nseq, n_prompt in cases:
    sess = tk.LlmSession(model, nseq, n_prompt + 8)
    toks = rng.integers(3, hp.vocab, (nseq, n_prompt)).astype(np.int32)
    first = sess.prefill(toks)  # captures the passes
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        again = sess.prefill(toks)
        best = min(best, time.perf_counter() - t0)
        assert np.array_equal(first, again)
    rows = nseq * n_prompt
    print("%d x %d tokens: %.2f ms per prefill, %.0f prompt tok/s, %.2f ms per 256-row pass, first ids %s  [%s]"
          % (nseq, n_prompt, 1e3 * best, rows / best, 1e3 * best / max(1, (rows + 255) // 256), first[:3].tolist(), form), flush=True)
    sess.close()
model.close()
