"""The two exact fp32 GEMMs against each other and against a float32 fma chain: the LDS-staged kernel (k_gemm_f32 / k_gemm_f32_big) and the
tiled kernel (csrc/nn/tk_gemm_tiled.hip) must agree bit for bit on every shape class the Whisper and fp16-LLM paths send them."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(gpu, M, N, K, act=0, f16=0, bias=True, residual=False, seed=0):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32) if bias else None
    r = rng.standard_normal((M, N)).astype(np.float32) if residual else None
    cs, ct = np.empty((M, N), np.float32), np.empty((M, N), np.float32)
    p = lambda x: x.ctypes.data_as(C.c_void_p) if x is not None else None
    gpu.check(gpu.lib().tk_mi355x_gemm_pair(0, M, N, K, p(a), p(w), p(b), p(r), act, f16, p(cs), p(ct)))
    return a, w, b, r, cs, ct


@pytest.mark.parametrize("M,N,K,act,bias,res", [
    (1, 16, 128, 0, False, False),        # one live row, one tile
    (32, 384, 384, 0, True, False),       # Whisper decoder linear, 32 utterances (2 M-tiles)
    (64, 1536, 384, 2, True, False),      # fc1 + GELU, 4 M-tiles
    (33, 51864 // 8, 384, 0, False, False),  # N not a multiple of 16 (6483): padded last tile
    (300, 384, 1536, 0, True, True),      # more than one 256-row block, residual epilogue
    (1500, 384, 1152, 2, True, False),    # conv2 as a GEMM, six row blocks, K = 9 * 128
    (257, 48, 128, 3, True, True),        # tails everywhere
])
def test_tiled_equals_staged_bitwise(gpu, M, N, K, act, bias, res):
    a, w, b, r, cs, ct = _pair(gpu, M, N, K, act, 0, bias, res, seed=M + N)
    assert np.array_equal(cs.view(np.uint32), ct.view(np.uint32))
    # a few outputs against the definition: one fma chain over k ascending from zero (float32 fma emulated in float64: exact for one step)
    rng = np.random.default_rng(1)
    for _ in range(8):
        i, j = int(rng.integers(M)), int(rng.integers(N))
        acc = np.float32(0)
        for k in range(K):
            acc = np.float32(np.float64(a[i, k]) * np.float64(w[j, k]) + np.float64(acc))
        if act == 0 and not res:
            want = np.float32(acc + (b[j] if bias else np.float32(0)))
            assert ct[i, j] == want


@pytest.mark.parametrize("M,N,K", [(16, 256, 1024), (48, 4096, 512), (256, 64, 256)])
def test_f16_weight_tiles_equal_the_staged_f16_path(gpu, M, N, K):
    """fp16-checkpoint semantics: f16 weights, activations rounded through f16, fp32 chain"""
    a, w, b, r, cs, ct = _pair(gpu, M, N, K, 0, 1, False, False, seed=7)
    assert np.array_equal(cs.view(np.uint32), ct.view(np.uint32))
    i, j = M - 1, N - 1
    acc = np.float32(0)
    for k in range(K):
        acc = np.float32(np.float64(np.float32(np.float16(a[i, k]))) * np.float64(np.float32(np.float16(w[j, k]))) + np.float64(acc))
    assert ct[i, j] == np.float32(acc + np.float32(0))


def test_gemm_pair_rejects_bad_shapes(gpu):
    z = np.zeros(4, np.float32).ctypes.data_as(C.c_void_p)
    assert gpu.lib().tk_mi355x_gemm_pair(0, 4, 4, 100, z, z, None, None, 0, 0, z, z) != 0   # K not a multiple of 128
    assert gpu.lib().tk_mi355x_gemm_pair(0, 0, 4, 128, z, z, None, None, 0, 0, z, z) != 0
    assert gpu.lib().tk_mi355x_gemm_pair(99, 4, 4, 128, z, z, None, None, 0, 0, z, z) != 0
