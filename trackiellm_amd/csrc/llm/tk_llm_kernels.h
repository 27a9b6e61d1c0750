/*
 * tk_llm_kernels.h — launch wrappers of the hand-written gfx950 kernels of the LLM stream.
 * Every wrapper only enqueues on `stream` (no allocation, no sync) so a whole decode step
 * can be captured in a hipGraph.
 */
#ifndef TK_LLM_KERNELS_H
#define TK_LLM_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tk_llm_layout.h"

struct TkGemvSeg {
    const uint8_t* tiles; /* device tiles [row_tile][K/256][tile bytes] */
    int type;             /* TK_TYPE_Q4_K / TK_TYPE_Q6_K */
    int row_tiles;        /* rows / 16; must be a multiple of 4 */
};

struct TkGemvArgs {
    TkGemvSeg seg[3];
    int nseg;
    int K;        /* reduction length, multiple of 256*ks */
    int ks;       /* K-split count: partial sums land in out[ks][16][n_total] */
    int n_total;  /* row pitch of `out`: the sum of rows over all segments of the matrix */
    int col0;     /* column of `out` where segment 0 starts (non-zero when a launch carries only some of the matrix's segments) */
    int nrows;    /* live rows (<= TK_MAX_ROWS); rows 16.. use the second M-tile */
    int swiglu;   /* 1 (only where tk_gemv_fuses_swiglu() says so): seg[0] = gate, seg[1] = up, out = h[nrows][n_total = d_ff] = silu(gate) * up */
    size_t aq_ts, ad_ts, abs_ts; /* M-tile strides of aq (bytes), ad (floats), abs (ints) */
    const int8_t* aq;
    const float* ad;
    const int8_t* abs; /* sub-block sums as (l, h) int8 images: [K/256][2][16][8] */
    const uint16_t* abs16; /* the same sums as two f16 images (sum = 2 hh + ll): [K/256][2][16][8] halves, M-tile stride abs_ts halves */
    float* out;
    /* fuse != 0 (only where tk_gemv_fuses_producer() says so: passes of one or two rows, where a launch boundary costs more than the
     * producer's work repeated in every workgroup): the launch builds its int8 activation image in LDS itself and aq / ad / abs are unused.
     *   1: residual update + RMS norm + Q8 — k_rmsnorm_q8's arithmetic: row = fx_in + (the fks slabs of fslab, ascending; pitch
     *      fn_total; fslab null: none), written once to fx_out (a buffer OTHER than fx_in: other workgroups still read it), norm
     *      weights fw, epsilon feps; K = d_model
     *   2: SwiGLU + Q8 — k_swiglu_q8's arithmetic on the fks slabs of the gate | up projection in fslab ([.][2 K] per row); K = d_ff */
    int fuse;
    const float* fx_in;
    float* fx_out;
    const float* fslab;
    int fks, fn_total;
    const float* fw;
    float feps;
};

struct TkActQ8 { /* quantised-activation buffers for one K */
    int8_t* aq;
    float* ad;
    int8_t* abs; /* (l, h) images of the sub-block sums, 256 B per 256-block */
    uint16_t* abs16; /* (hh, ll) f16 images of the same sums, 512 B per 256-block: the batched kernel's one-MFMA min term */
    size_t aq_ts, ad_ts, abs_ts; /* M-tile strides */
    /* f16-weight matrices (fp16 checkpoints) consume the activations as f32 values rounded through f16 — what a CPU engine's f16 matmul
     * does to its f32 input; null when the model has no such matrix */
    float* af;    /* the tiled GEMM's operand image (csrc/nn/tk_gemm_tiled.h): [M-tile][K / 16][4 g][16 rows][4 t] floats, k = 16 j + 4 t + g */
    size_t af_ts; /* floats between M-tiles = 16 K */
};


/* weights */
void tk_launch_synth_blocks(int type, uint64_t seed, uint64_t tensor_id, int64_t nblocks, float scale, void* out, hipStream_t s);
void tk_launch_synth_f32(uint64_t seed, uint64_t tensor_id, int64_t n, float* out, hipStream_t s);
/* W += scale (B A) on a matrix in GGUF layout (Q4_K / Q6_K blocks or f16), quantised back to its own type; A [r][K], B [rows][r] on the device.
 * false: arguments the kernel does not take (type, K % 256, more than 2^31 blocks) */
bool tk_launch_lora_merge(int type, void* blocks, int64_t rows, int64_t K, const float* A, const float* B, int r, float scale, hipStream_t s);
void tk_launch_repack(int type, const void* blocks, int64_t rows, int64_t K, uint8_t* tiles, hipStream_t s);

/* step kernels */
void tk_launch_embed(const void* embd, int type /* TK_TYPE_Q4_K or TK_TYPE_F16 */, int D, const int32_t* tok, int nrows, float* x, hipStream_t s);
void tk_launch_synth_f16(uint64_t seed, uint64_t tensor_id, int64_t n, float scale, uint16_t* out, hipStream_t s);
void tk_launch_rmsnorm_q8(float* x, const float* partial, int ks, int n_total_partial, const float* w, float eps, int D, int nrows,
                          TkActQ8 out, hipStream_t s);
void tk_launch_residual_fold(float* x, const float* partial, int ks, int n_total, int D, int nrows, hipStream_t s);
void tk_launch_gemv(const TkGemvArgs& a, hipStream_t s);
void tk_launch_qkv_rope_append(const float* partial, int ks, int n_total, int n_head, int n_kv_head, int head_dim, const float* rope_cos,
                               const float* rope_sin, const int32_t* seq, const int32_t* pos, int nrows, float* qbuf, uint16_t* kcache,
                               uint16_t* vcache, int layer, int max_seq, int max_ctx, hipStream_t s);
void tk_launch_attention(const float* qbuf, const float* partial, int ks, int n_total, const float* rope_cos, const float* rope_sin,
                         uint16_t* kcache, uint16_t* vcache, const int32_t* seq, const int32_t* pos, int nrows, int n_head, int n_kv_head,
                         int head_dim, int layer, int max_seq, int max_ctx, TkActQ8 out, bool fused, hipStream_t s);
void tk_launch_swiglu_q8(const float* partial, int ks, int FF, int nrows, TkActQ8 out, hipStream_t s);
/* wide passes: the gate | up launch forms h = silu(gate) * up in its epilogue (half the slab bytes written and read back); then only the quantisation is left */
/* rows up to which the norm / SwiGLU producer of a mat-vec launch runs inside it (TkGemvArgs::fuse) */
#define TK_GEMV_FUSE_MAX_ROWS 2
bool tk_gemv_fuses_producer(int nrows, int K, int ks, int fks);
bool tk_gemv_fuses_swiglu(int nrows, int ks, int type_gate, int type_up);
void tk_launch_quant_q8(const float* hbuf, int FF, int nrows, TkActQ8 out, hipStream_t s);
#include "../common/tk_sample.h"
/* allow_base / allow_row (both optional): per-row allowed-token bit masks, allow_row[r] = mask index or -1; samp (optional): [nrows] */
void tk_launch_argmax(const float* logits, int vocab, int nrows, const uint32_t* allow_base, const int32_t* allow_row, TkSampleRow* samp, int32_t* tok,
                      int32_t* pos, int32_t* nsteps, int32_t* hist, int hist_stride, hipStream_t s);

/* the attention launch a pass takes: kernel 0 = k_attention<gq, fused, head_dim, chunk, slots> (chunk = positions per ring slot), kernel 1 =
 * k_attention_narrow (gq 2, chunk = positions resident per chunk, the whole context when it fits) */
struct TkAttentionPlan { int kernel, gq, chunk, slots; size_t lds_bytes; };
/* a decode session on `device` came (+1) or went (-1): with more than one alive the plan prefers forms that share a CU with other streams' launches */
void tk_attention_note_session(int device, int delta);
TkAttentionPlan tk_attention_plan(int nrows, int n_head, int n_kv_head, int head_dim, int max_ctx, bool fused);

/* multi-position passes (prompt chunks): 16 rows of a sequence per workgroup on the fp32 matrix pipe, bit-identical to k_attention's non-fused
 * form.  tiles: 1 + TK_MAX_ROWS ints built by tk_launch_att_tiles from the pass's sequence ids (once per pass); the cache must already hold
 * the pass's own K / V rows (tk_launch_qkv_rope_append) */
bool tk_attention_prefill_applies(int n_head, int n_kv_head, int head_dim);
void tk_launch_att_tiles(const int32_t* seq, int nrows, int32_t* tiles, hipStream_t s);
void tk_launch_attention_prefill(const float* qbuf, const uint16_t* kcache, const uint16_t* vcache, const int32_t* seq, const int32_t* pos,
                                 const int32_t* tiles, int nrows, int n_head, int n_kv_head, int head_dim, int layer, int max_seq, int max_ctx, TkActQ8 out,
                                 hipStream_t s);
/* decode passes of few rows over long contexts: scores spread over (row, KV head, 64-position block) workgroups, then one sequential PV chain per
 * (row, head, class) wave — bit-identical to the fused kernels.  scores: tk_attention_long_scratch_floats() floats; `partial` holds the
 * q | k | v projection's K-split slabs (the launch finishes q / k / v itself and appends this pass's rows to the cache) */
/* by measurement, end to end (profiles/r05_attention_long_ctx.txt): 1 .. 4 rows from position 512, 5 .. 8 rows from 768 (the isolated launch
 * already wins from ~384 at one row; a whole step does not before ~500); at 16 rows the fused kernels' 256 workgroups already cover the chip */
#define TK_LONG_ATT_MAX_ROWS 8
#define TK_LONG_ATT_MIN_POS 512
static inline int tk_long_att_min_pos(int nrows) { return nrows > 4 ? 768 : TK_LONG_ATT_MIN_POS; }
bool tk_attention_long_applies(int nrows, int n_head, int n_kv_head, int head_dim);
size_t tk_attention_long_scratch_floats(int n_head, int head_dim, int max_ctx); /* floats of the `scores` scratch buffer */
void tk_launch_attention_long(const float* partial, int ks, int n_total, const float* rope_cos, const float* rope_sin, uint16_t* kcache, uint16_t* vcache,
                              const int32_t* seq, const int32_t* pos, int nrows, int n_head, int n_kv_head, int head_dim, int layer, int max_seq, int max_ctx,
                              float* scores, TkActQ8 out, hipStream_t s);
size_t tk_gemv_lds_bytes(int K, int ks, int mtiles);
/* dynamic LDS of one k_attention workgroup; must stay below 160 KiB (the session checks it against its max_ctx) */
size_t tk_attention_lds_bytes(int gq, int head_dim, int max_ctx, int chunk /* positions per ring slot: 32 or 64 */, int slots = 2 /* ring depth: 2 or 5 */);
/* opts every kernel of this file into 160 KiB of dynamic LDS on `device` (which must be the calling thread's current device); idempotent,
 * thread-safe; returns nullptr or an error string.  Sessions call it at creation: launches never change function attributes. */
const char* tk_llm_prepare_device(int device);

#endif
