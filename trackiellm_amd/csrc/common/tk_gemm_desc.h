/* tk_gemm_desc.h — descriptor of the one fp32 GEMM both the HIP kernels and the oracle evaluate (no HIP types). */
#ifndef TK_GEMM_DESC_H
#define TK_GEMM_DESC_H

#include <stdint.h>

enum TkAct { TK_ACT_NONE = 0, TK_ACT_SILU = 1, TK_ACT_GELU = 2, TK_ACT_SIGMOID = 3 };

struct TkGemm {
    const float* A; /* [M][lda]  (k contiguous) */
    const float* B; /* b_kn == 0: [N][ldb] (k contiguous);  b_kn == 1: [K][ldb] (n contiguous) */
    float* C;       /* [M][ldc] */
    const float* bias;     /* [N] or null */
    const float* residual; /* [M][ldr] or null; added after the activation */
    int M, N, K, lda, ldb, ldc, ldr;
    int b_kn;
    int act;
    float alpha; /* out = act(alpha * acc + bias); alpha == 1 is skipped exactly */
    int batch;
    int64_t sA, sB, sC, sR; /* batch strides in floats */
    /* optional second batch level: z = zo * batch_inner + zi ; offset = zo * s?2 + zi * s? (batch_inner == 0: single level) */
    int batch_inner;
    int64_t sA2, sB2, sC2, sR2;
    /* B holds IEEE f16 values (2 bytes each, same [N][ldb] / [K][ldb] indexing in elements): each is widened to f32 exactly, the chain
     * arithmetic is unchanged — f16 checkpoints (LLM fp16 weights) stream half the bytes */
    int b_f16;
    /* hint, no arithmetic: C is only read as the A operand of the next linear layer (fc1 -> fc2); a backend may emit it pre-packed */
    int c_feeds_linear;
    /* implicit im2col (im_C > 0): A is a convolution's NHWC input [B][im_H][im_W] pixels of pitch im_ldx; GEMM row m = (b, oy, ox) over
     * im_Ho x im_Wo outputs and column k = (ky im_kw + kx) im_C + c address it directly (zero outside the image), K = kh kw im_C.
     * Same values in the same order as the explicit column matrix; lda, batch and b_kn are unused */
    int im_C, im_H, im_W, im_ldx, im_kw, im_stride, im_pad, im_Ho, im_Wo;
    /* opt-in (0 = the exact chain above, always what the oracle evaluates): the HIP backend may contract on the f16 matrix pipe with every
     * operand split into two f16 halves (x = hi + lo / 2048 to ~22 bits; hi.hi + (hi.lo + lo.hi) / 2048 with fp32 accumulation): NOT the
     * chain's bits — within ~1e-6 of its scale — for callers that asked for speed under a tolerance (csrc/nn/tk_nn_kernels.hip: k_gemm_h3_*) */
    int fast;
};


#endif
