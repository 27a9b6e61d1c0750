#!/usr/bin/env python3
"""One runner through the reference entry points (developer tool, needs an MI355X): tk_model_loader_load_model ->
tk_llm_runner_prepare_generation -> N x tk_llm_runner_generate_next_token on the synthetic Mistral-7B Q4_K_M, ms per token.
TK_MI355X_NO_FUSE=1 keeps the norm / SwiGLU producers as launches of their own (A/B); under rocprofv3 --kernel-trace --stats the per-kernel
durations of the decode step come out (ROC_AQL_QUEUE_SIZE=524288 for the graph path, DESIGN.md "Profiling")."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trackiellm_amd as tk  # noqa: E402

N = int(os.environ.get("TK_B1_TOKENS", "128"))
loader = tk.ModelLoader()
h = loader.load("synthetic://mistral-7b?seed=4")
runner = tk.LlmRunner(h, context_size=512)
runner.prepare("The user is in a room. " * 8)
for _ in range(8):
    runner.next_token()
t = time.time()
n = 0
for _ in range(N):
    if runner.next_token() is None:
        break
    n += 1
dt = time.time() - t
print(f"one runner: {n} tokens, {1000 * dt / max(n, 1):.3f} ms per token, {n / dt:.1f} tok/s (TK_MI355X_NO_FUSE={os.environ.get('TK_MI355X_NO_FUSE', '0')})", flush=True)
runner.close()
loader.unload(h)
loader.close()
