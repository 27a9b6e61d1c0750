"""softmax_4x256.npz — the reference's only numeric known-answer test, tests/tk_gpu_softmax_test.cpp:24-69: a 4 x 256 f32 tensor filled
with i % 256, row softmax computed on the CPU the way that test's checker does (f32 max, f32 running sum of expf(x - max) in column
order, expf(x - max) / sum), compared with the GPU at 1e-6 absolute.  libm's expf is called through ctypes so the expected values are
the ones the reference's C++ checker would produce on this platform.  Run:  python tests/golden/make_softmax_golden.py"""
import ctypes
import ctypes.util
import os

import numpy as np

ROWS, COLS = 4, 256
libm = ctypes.CDLL(ctypes.util.find_library("m"))
libm.expf.restype = ctypes.c_float
libm.expf.argtypes = [ctypes.c_float]

x = (np.arange(ROWS * COLS) % COLS).astype(np.float32).reshape(ROWS, COLS)
out = np.empty_like(x)
for r in range(ROWS):
    m = np.float32(x[r, 0])
    for j in range(1, COLS):
        if x[r, j] > m:
            m = x[r, j]
    s = np.float32(0.0)
    for j in range(COLS):
        s = np.float32(s + np.float32(libm.expf(float(np.float32(x[r, j] - m)))))
    for j in range(COLS):
        out[r, j] = np.float32(np.float32(libm.expf(float(np.float32(x[r, j] - m)))) / s)
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "softmax_4x256.npz"), input=x, expected=out, tolerance=np.float32(1e-6))
print("rows sum to", out.sum(1), "max", out.max())
