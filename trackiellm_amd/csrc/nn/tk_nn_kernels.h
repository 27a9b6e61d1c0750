/*
 * tk_nn_kernels.h — fp32 building blocks shared by the detector and ASR streams.
 *
 * The reference runs these networks inside ONNX Runtime / whisper.cpp (absent: SURVEY.md §0 F1).
 * Here every dense contraction (conv as implicit GEMM, linear layers, QK^T, PV, the DFT and
 * mel filterbank) goes through ONE kernel, tk_gemm_f32, built on v_mfma_f32_32x32x2_f32.
 * That instruction is bit-for-bit a k-ordered fp32 fma chain, so with the accumulator started at
 * zero and k walked in ascending order the result equals the oracle's
 *     acc = 0; for k: acc = fmaf(a[k], b[k], acc);  out = act(acc + bias) (+ residual)
 * exactly — the detector and ASR parity tests are 0-ulp, like the LLM ones.
 * fp32 (not bf16) is deliberate: YOLOv8n is 8.7 GFLOP and the Whisper-tiny encoder 37 GFLOP per
 * call, ~1 ms of fp32 MFMA against a >= 70 ms LLM decode; exactness is worth more than speed here.
 */
#ifndef TK_NN_KERNELS_H
#define TK_NN_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../common/tk_gemm_desc.h"

void tk_launch_gemm(const TkGemm& g, hipStream_t s);
/* whether a convolution can be addressed by the GEMM itself (TkGemm::im_*): channels and pitch multiples of 4, input x and the [N][K]
 * f32 weights w 16-byte aligned */
bool tk_gemm_im2col_ok(const float* x, int C, int ldx, const float* w);
/* direct 3x3 convolution of a 3-channel image into 16 channels (the detector's stem), the GEMM's chain and epilogue (bias, act) per
 * output; false (nothing launched): another shape — use im2col + tk_launch_gemm */
bool tk_launch_conv_stem(const float* x, int B, int H, int W, int C, int ldx, const float* w, const float* bias, int act, int N, int k, int stride, int pad, float* y,
                         int ldy, hipStream_t s);
/* opts the large-tile GEMM into its dynamic LDS on the calling thread's current device; idempotent, thread-safe.  tk_launch_gemm does
 * it on first use; callers that capture launches into a hipGraph call it beforehand. */
bool tk_nn_prepare_device();
/* opt-in (the fast contraction's companion, TkGemm::fast): attention over Tk keys for Tq queries per (sequence, head) in ONE kernel — Q K^T, online
 * softmax, P V on the f16 matrix pipe with split operands, no score matrix in HBM; head_dim 64, q / k / v / out rows of pitch d = nh * 64,
 * 16-byte aligned.  false (nothing launched): another geometry.  ~1e-6 of scale off the exact three-launch form, not its bits. */
bool tk_launch_attention_h3(const float* q, const float* k, const float* v, float* out, int B, int nh, int Tq, int Tk, int64_t q_bstride, int64_t kv_bstride, int d,
                            float scale, hipStream_t s);

/* NHWC im2col: col[b*Ho*Wo + oy*Wo + ox][(ky*kw + kx)*C + c]; input row stride ldx floats per pixel */
void tk_launch_im2col(const float* x, int B, int H, int W, int C, int ldx, int kh, int kw, int stride, int pad, float* col, hipStream_t s);
/* 1-D variant for conv1d over [T][C] rows */
void tk_launch_maxpool5(const float* x, int B, int H, int W, int C, int ldx, float* y, int ldy, hipStream_t s);
void tk_launch_upsample2x(const float* x, int B, int H, int W, int C, int ldx, float* y, int ldy, hipStream_t s);
void tk_launch_copy_cols(const float* x, int rows, int C, int ldx, float* y, int ldy, hipStream_t s);
/* rows of [T][C] (row pitch ldx) -> col[b*To + t][(kx*C + c)], kernel width kw, zero padding `pad` on both ends */
void tk_launch_im2col1d(const float* x, int B, int T, int C, int ldx, int kw, int stride, int pad, float* col, hipStream_t s);
/* out[r][:] = table[idx[r]][:] + pos[(pos0[r])][:] */
void tk_launch_embed_rows(const float* table, const float* pos, const int32_t* idx, const int32_t* pos_idx, int rows, int D, float* out, hipStream_t s);
void tk_launch_argmax_rows(const float* x, int rows, int cols, int ld, int32_t* out, hipStream_t s);
/* token pick under whisper.cpp's decoding policy: arg max (temp 0) or one draw from softmax(l / temp), + the pick's log-probability under
 * that distribution per row (logprob may be null); the generator is keyed by (seed, counter0 + row) */
struct TkPick { float temp; uint64_t seed; uint32_t counter0; float* logprob; };
void tk_launch_pick_rows(const float* x, int rows, int cols, int ld, int32_t* out, const TkPick& pk, hipStream_t s);
/* whisper.cpp's whisper_process_logits + the per-token bookkeeping of whisper_full's decode loop, as the reference's wrapper configures them
 * (src/audio/tk_asr_whisper.c:89-110: suppress_blank off, suppress_non_speech_tokens on, timestamps on, max_initial_ts at its default 1.0 s):
 * per row a static suppression table, the timestamp rules (pairs, initial cap, monotonic), "timestamps win when their summed probability beats
 * every text token", then the pick (arg max or one draw) and its log-probability, then the row's decode state moves on.
 * state[row][8] = {tokens sampled, last was a timestamp, the one before was, has_ts, seek_delta, result_len, status (0 running, 1 completed,
 * 2 failed), seek_end (10 ms frames of the utterance)}; a row whose status is not 0 emits `eot` and stands still. */
#define TK_WH_STATE_INTS 8
struct TkWhFilter { const uint8_t* suppress; int32_t* state; int32_t beg, eot, tid0; };
void tk_launch_pick_rows_filtered(const float* x, int rows, int cols, int ld, int32_t* out, const TkPick& pk, const TkWhFilter& f, hipStream_t s);
void tk_launch_layernorm(const float* x, int rows, int D, const float* w, const float* b, float eps, float* y, hipStream_t s, float* img = nullptr /* optional: also the tiled GEMM's operand image of y */);
void tk_launch_softmax_rows(float* x, int rows, int cols, int ld, hipStream_t s);
void tk_launch_add_rows(float* x, const float* add, int rows, int D, int add_rows, hipStream_t s);

#endif
