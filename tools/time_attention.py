#!/usr/bin/env python3
"""Isolated k_attention timing over (rows per pass, cached positions) — developer tool, needs an MI355X.  TK_MI355X_LIB selects a variant build."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trackiellm_amd as tk  # noqa: E402

hp = tk.MISTRAL_7B()
hp.n_layer = 2
model = tk.LlmModel(hp).fill_synthetic(4)
max_ctx = int(os.environ.get("TK_ATT_MAX_CTX", "520"))  # the session's context capacity sizes the kernel's score buffer (LDS -> workgroups per CU)
sess = tk.LlmSession(model, 256, max_ctx)
out = []
cases = [(256, 64), (256, 128), (256, 192), (256, 500), (16, 128), (16, 500), (1, 128)]
if os.environ.get("TK_ATT_CASES"):  # "rows:ctx,rows:ctx,..."
    cases = [tuple(int(v) for v in c.split(":")) for c in os.environ["TK_ATT_CASES"].split(",")]
elif len(sys.argv) > 2:  # tools/time_attention.py ROWS CTX: one case (the PMC passes of tools/collect_profiles.sh)
    cases = [(int(sys.argv[1]), int(sys.argv[2]))]
for rows, ctx in cases:
    ms, kvb = sess.time_attention(rows, ctx, 64)
    out.append(f"{rows}x{ctx}: {1000 * ms:.1f} us ({kvb / ms / 1e6:.0f} GB/s)")
print(os.environ.get("TK_MI355X_LIB", "default").split("/")[-1], " | ".join(out), flush=True)
