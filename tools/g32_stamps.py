#!/usr/bin/env python3
"""diagnostic: phase durations (shader cycles) of k_gemm32_w4a8 from the s_memtime stamps of a -DTK_G32_STAMPS build
   TK_MI355X_LIB=build/variants/libtrackie_stamps.so python3 tools/g32_stamps.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trackiellm_amd as tk
model = tk.LlmModel(tk.MISTRAL_7B(), device=0).fill_synthetic(4)
sess = tk.LlmSession(model, 256, 64)
ms, b = sess.time_gemv(0, 0, 256, 3)   # gate_up, 16 blocks
st = np.zeros((2, 16, 16), np.uint64)
assert tk.lib().tk_mi355x_debug_stamps(st.ctypes.data_as(C.c_void_p)) == 0
print("gate_up launch", ms * 1e3, "us")
t0 = int(st[0, 0, 15])
names = ["barrier_out", "unpack_done"] + [f"t{t}_{p}" for t in range(4) for p in ("A_in", "mfma_issued", "finished")]
for h in range(2):
    print(f"--- wave {4*h} (row half {h}): per block, cycles since previous stamp: top=loop top->vm wait, bar=wait->barrier out, unp=unpack, then per tile A-wait / MFMA-issue / finish")
    for b in range(16):
        s = st[h, b].astype(np.int64)
        seq = [s[15], s[14], s[0], s[1]] + [s[2 + i] for i in range(12)]
        d = [int(seq[i + 1] - seq[i]) for i in range(len(seq) - 1)]
        print(f"b{b:2d} start {int(s[15]) - t0:7d}: vmwait {d[0]:5d} bar {d[1]:5d} unp {d[2]:5d} | " + " | ".join(f"{d[3+3*t]:4d} {d[4+3*t]:4d} {d[5+3*t]:4d}" for t in range(4)))
