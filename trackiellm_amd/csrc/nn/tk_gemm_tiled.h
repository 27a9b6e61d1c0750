/*
 * tk_gemm_tiled.h — C[rows][N] = A[rows][K] x W[N][K]^T on the exact fp32 MFMA (v_mfma_f32_16x16x4_f32: an fma chain over its four k,
 * bitwise) with BOTH operands laid out for the matrix cores ahead of time:
 *   W (a model's weights, constant): tiled once at load time — per 16 output columns and 32 k one contiguous piece, 64 lanes x 8 values;
 *     lane (n = l % 16, g = l / 16) holds k = 32 c + 4 t + g, t = 0 .. 7.  f16 values (16 B per lane: fp16 LLM checkpoints) or f32 values
 *     (32 B per lane: Whisper linears).  Rows beyond N in the last tile are zero.
 *   A (activations): an operand image [M-tile of 16 rows][K / 16][4 g][16 rows][4 t] floats, element (row, k) with k = 16 j + 4 t + g.
 *     Producers either write it directly (the LLM's norm / activation kernels, csrc/llm/tk_llm_kernels.hip: quantize_chunk8) or
 *     tk_launch_pack_a() converts a row-major matrix.
 * Contract: per K-split slab one fp32 chain over k ascending from zero (the same chain k_gemm_f32 evaluates on the 32x32x2 MFMA, so the
 * two kernels agree bit for bit).  ks > 1: out[ks][TK_MAX_ROWS rows][ldc] partial slabs, summed in ascending order by the consumer
 * (LLM path); ks == 1: out[row][ldc] with the optional epilogue  v = act(acc + bias[n]) + residual[row][n].
 *
 * A wave owns one weight row tile of one K range and streams it as one piece per 32 k, PF pieces in flight in registers; a piece feeds
 * 8 MFMAs per M-tile.  The A image is staged through an LDS ring by LDS-DMA, 1 KiB per (16 k, M-tile), shared by the workgroup's
 * waves; a lane's operand of four consecutive MFMAs is one ds_read_b128.  Rows beyond 256 run as further workgroups (grid y) over the
 * same weights.  At 16 rows the f16 path is HBM-bound (2 B per weight); from 32 rows up the fp32 MFMA rate (157 TFLOP/s) bounds it.
 */
#ifndef TK_GEMM_TILED_H
#define TK_GEMM_TILED_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../common/tk_gemm_desc.h"

#define TK_TW_ROWS_PER_TILE 16
#define TK_TW_MAX_BLOCK_ROWS 256 /* rows one workgroup covers: 16 M-tiles */

struct TkTiledGemm {
    const uint8_t* tiles[3]; /* up to three weight matrices side by side in N (q | k | v) */
    int row_tiles[3];
    int nseg;
    int wbytes;              /* 2: f16 tiles, 4: f32 tiles */
    int K, ks;
    int ldc;                 /* row pitch of out (and of the K-split slabs) */
    int n_valid;             /* columns actually stored (N; the last tile may be padding) */
    int nrows;
    int slab_rows;           /* rows per K-split slab (ks > 1): TK_MAX_ROWS of the LLM path */
    const float* a_img;
    size_t a_ts;             /* floats between M-tiles of the image = 16 K */
    float* out;
    const float* bias;       /* ks == 1 only from here on */
    const float* residual;
    int ldr;
    int act;                 /* TkAct */
    int add_zero_bias;       /* k_gemm_f32 adds 0.0f when there is no bias (-0 + 0 = +0): reproduce it where results must match that kernel */
    /* ks == 1 extras (zero = off):
     * per_seg: every segment is a linear layer of its own on the same input — its own destination, pitch, bias and width (q | k | v of a
     *   Whisper decoder step in ONE launch: q goes to the query buffer, k and v straight into row p of the caches) */
    int per_seg;
    float* seg_out[3];
    int seg_ldc[3];
    const float* seg_bias[3];
    int seg_n[3];
    /* c_img: the output is ALSO written as the operand image of the next linear layer (K = n_valid), so no pack launch follows */
    float* c_img;
};

/* false (nothing launched): K / ks is not a multiple of the ring granularity (128 k; 64 k from 129 rows per block on) or a field is inconsistent */
bool tk_launch_gemm_tiled(const TkTiledGemm& g, hipStream_t s);
/* row-major f16 / f32 [N][K] (device) -> tiles; N is padded up to a multiple of 16 with zero rows, K must be a multiple of 32 */
size_t tk_tiled_weight_bytes(int64_t N, int64_t K, int wbytes);
void tk_launch_tile_weights(const void* src, int wbytes, int64_t N, int64_t K, uint8_t* tiles, hipStream_t s);
/* row-major A [rows][lda] (device) -> operand image; round_f16: values rounded through f16 on the way (fp16-checkpoint semantics) */
size_t tk_a_image_floats(int64_t rows, int64_t K);
void tk_launch_pack_a(const float* A, int64_t rows, int K, int lda, int round_f16, float* img, hipStream_t s);
/* opts the kernels into their dynamic LDS on the calling thread's current device; idempotent, thread-safe */
bool tk_gemm_tiled_prepare_device();

#endif
