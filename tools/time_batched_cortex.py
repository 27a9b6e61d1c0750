#!/usr/bin/env python3
"""K cortex handles on one model file, one data-dependent cycle each through tk_cortex_* only (bench.py: reference_abi_batched_cortex) —
developer tool, needs an MI355X.    python tools/time_batched_cortex.py 16,64,256 [tokens per cycle] [--progress]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import trackiellm_amd as tk  # noqa: E402
import bench  # noqa: E402

progress = "--progress" in sys.argv
argv = [a for a in sys.argv[1:] if a != "--progress"]
ks = [int(v) for v in (argv[0] if argv else "16,64").split(",")]
N = int(argv[1]) if len(argv) > 1 else 128
for K in ks:
    print(json.dumps(bench.reference_abi_batched_cortex(tk, K, N, progress=progress)), flush=True)
