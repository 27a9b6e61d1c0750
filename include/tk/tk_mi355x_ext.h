/*
 * tk_mi355x_ext.h — extensions of the drop-in library that have no reference counterpart.
 *
 * The reference runs ONE sequence per runner on one thread (src/cortex/tk_cortex_main.c:957-994).
 * SURVEY.md §0 F9 shows the headline target needs >= 6 cortex cycles decoded concurrently so the
 * 4.3 GB weight stream is shared; these entry points expose that batched path (and the pieces the
 * parity tests drive directly) with plain pointers and sizes only.
 */
#ifndef TK_MI355X_EXT_H
#define TK_MI355X_EXT_H

#include "tk_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t n_layer, d_model, n_head, n_kv_head, head_dim, d_ff, vocab;
    float rms_eps, rope_theta;
    int32_t ks_qkv, ks_o, ks_gateup, ks_down, ks_out; /* K-split plan (canonical summation order) */
} tk_mi355x_llm_hparams_t;

typedef struct tk_mi355x_llm_model_s tk_mi355x_llm_model_t;
typedef struct tk_mi355x_llm_session_s tk_mi355x_llm_session_t;

TK_API const char* tk_mi355x_version(void);
TK_API int tk_mi355x_device_count(void);
/* HIP device used by the reference entry points whose config carries no device id (tk_vad_silero_create,
 * tk_asr_whisper_create, tk_preprocessor_*, tk_model_loader_load_model); default 0.  One process per GPU sets its rank's device. */
TK_API void tk_mi355x_set_default_device(int device);
TK_API int tk_mi355x_get_default_device(void);

/* models ----------------------------------------------------------------------------------- */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_llm_model_create(tk_mi355x_llm_model_t** out, const tk_mi355x_llm_hparams_t* hp, int device);
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_llm_model_fill_synthetic(tk_mi355x_llm_model_t* m, uint64_t seed);
/* the fp16 checkpoint recipe (BASELINE configs[4]): every matrix and the token embedding IEEE f16 (GGUF type 1), norms f32.  f16 matrices run
 * on the exact fp32 MFMA GEMM over f16-rounded activations (one k-ordered chain per output, no K-split): results stay bit-identical to
 * the oracle.  Loader name: synthetic://mistral-7b-f16, synthetic://tiny-f16; GGUF files with F16 tensors load the same way. */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_llm_model_fill_synthetic_f16(tk_mi355x_llm_model_t* m, uint64_t seed);
/* tensor in GGUF block layout; layer = -1 for {0 token_embd, 1 output_norm, 2 output}, else
 * {0 attn_norm,1 q,2 k,3 v,4 o,5 ffn_norm,6 gate,7 up,8 down}; type = ggml type id (0 F32, 1 F16, 12 Q4_K, 14 Q6_K) */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_llm_model_set_tensor(tk_mi355x_llm_model_t* m, int layer, int which, int type, const void* data,
                                                                   size_t nbytes);
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_llm_model_load_gguf(tk_mi355x_llm_model_t** out, const char* path, int device);
/* LoRA adapters.  The reference applies one right after the model is loaded, in place and once: llama_model_apply_lora_from_file
 * (src/ai_models/tk_model_loader.c:259-270) — W' = W + (alpha / r) B A, a quantised W dequantised, added to and quantised back to its own type.
 * Here the same merge runs on the GPU while a matrix is installed, so an adapted model decodes at the un-adapted model's speed.
 * tk_model_load_params_t.lora_adapter is the reference's way in; these are the pieces:
 *   load_gguf_lora: load_gguf with an adapter (NULL / "" = none);
 *   set_lora: the adapter to merge into every matrix installed FROM NOW ON (set_tensor, fill_synthetic); NULL / "" = none.  Fails with
 *     TK_ERROR_MODEL_LOAD_FAILED (the reference's code) on an unreadable file or factor shapes that do not fit the model;
 *   lora_merged: matrices the adapter has changed so far;
 *   lora_probe: rank, alpha and tensor-pair count of an adapter file ("ggla" v1 — what llama_model_apply_lora_from_file read — or a GGUF
 *     adapter, general.type = "adapter"); no GPU involved. */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_llm_model_load_gguf_lora(tk_mi355x_llm_model_t** out, const char* path, const char* lora_path, int device);
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_llm_model_set_lora(tk_mi355x_llm_model_t* m, const char* adapter_path);
TK_API int tk_mi355x_llm_model_lora_merged(const tk_mi355x_llm_model_t* m);
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_lora_probe(const char* path, int32_t* rank, float* alpha, int32_t* n_tensors);
/* parses GGUF metadata only (runs without a GPU); n_vocab_tokens optional */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_gguf_probe(const char* path, tk_mi355x_llm_hparams_t* out, int32_t* n_vocab_tokens);
/* token ids of `text` under the vocabulary of a GGUF file (CPU only); returns the count (may exceed cap) or -1 */
TK_API int tk_mi355x_gguf_tokenize(const char* path, const char* text, int add_bos, int32_t* ids, int cap);
TK_API void tk_mi355x_llm_model_get_hparams(const tk_mi355x_llm_model_t* m, tk_mi355x_llm_hparams_t* out);
TK_API uint64_t tk_mi355x_llm_model_weight_bytes(const tk_mi355x_llm_model_t* m);
TK_API void tk_mi355x_llm_model_destroy(tk_mi355x_llm_model_t** m);

/* continuous batching behind tk_llm_runner_* (csrc/llm/tk_llm_batcher.h): the runners created on one model handle (what
 * tk_model_loader_load_model returns) share decode sessions of `slots` sequences; their prepare_generation / generate_next_token rows
 * are coalesced into passes by one scheduler thread per session.  Set before the first tk_llm_runner_create on the model (default:
 * $TK_MI355X_RUNNER_SLOTS, else 16; at most tk_mi355x_llm_max_rows()).  KV memory = slots x context_size x 128 KiB for Mistral-7B. */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_llm_model_set_runner_slots(void* model_handle, int slots);
/* Passes of one or two rows (one runner stepping alone) run their RMS-norm and SwiGLU producers inside the mat-vec launches that consume
 * them (five launches per layer instead of eight; same values); $TK_MI355X_NO_FUSE=1, read when a pass is recorded, keeps them as launches of
 * their own.  $TK_MI355X_NO_GRAPH=1 launches every pass eagerly instead of replaying hipGraphs (profilers).
 * Passes that hold several positions of a sequence (prompt chunks) and reach position 128 run their attention with 16 rows of a sequence per
 * workgroup on the fp32 matrix pipe (k_attention_prefill; same values as the per-row kernel): $TK_MI355X_NO_PREFILL_ATT=1 keeps the per-row
 * kernel everywhere.  $TK_MI355X_TIME_HOT=1 makes the stand-alone mat-vec timing (tk_mi355x_llm_time_gemv) repeat ONE layer's launch — weights
 * resident in the caches — instead of cycling through the layers.  $TK_MI355X_ASR_POLICY=1: include/tk/tk_audio.h. */
/* what the schedulers of a model have done so far: passes run, rows processed, the widest pass */
TK_API void tk_mi355x_llm_model_batch_stats(void* model_handle, uint64_t* passes, uint64_t* rows, int32_t* max_rows_in_a_pass);
/* run-ahead rows (csrc/llm/tk_llm_batcher.h) that no owner came back for: the scheduler feeds a sequence's sampled id one position ahead
 * of its owner's next tk_llm_runner_generate_next_token; a runner that stops or changes course costs one such row.  `rows` above counts
 * only rows an owner asked for. */
TK_API uint64_t tk_mi355x_llm_model_run_ahead_wasted(void* model_handle);

/* sessions --------------------------------------------------------------------------------- */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_llm_session_create(tk_mi355x_llm_session_t** out, tk_mi355x_llm_model_t* m, int max_seq,
                                                                 int max_ctx);
TK_API void tk_mi355x_llm_session_destroy(tk_mi355x_llm_session_t** s);
/* rows one pass can hold (16-row MFMA M-tiles x 8): the row capacity of forward(), the column count of decode()'s out_tokens */
TK_API int tk_mi355x_llm_max_rows(void);
/* one pass over nrows <= tk_mi355x_llm_max_rows() (sequence, position, token) rows; logits [nrows][vocab] and argmax [nrows] optional */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_llm_forward(tk_mi355x_llm_session_t* s, int nrows, const int32_t* seq, const int32_t* pos,
                                                          const int32_t* tok, float* logits, int32_t* argmax);
/* one pipeline stage of a pass (the LLM layer-sharded over GPUs): layers [layer0, layer1) of this rank's model.  The
 * first stage starts from tok (x_in NULL), later stages from the residual stream x_in [nrows][d_model] fp32; every stage but the last
 * writes the stream to x_out; the last (head != 0, layer1 == n_layer) samples into argmax[nrows].  x_on_host != 0: x_in / x_out are
 * host pointers (gloo), else device pointers on this session's GPU (RCCL send / recv buffers).  Synchronous.  Bit-identical to
 * tk_mi355x_llm_forward whatever the split. */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_llm_forward_stage(tk_mi355x_llm_session_t* s, int nrows, const int32_t* seq, const int32_t* pos,
                                                                const int32_t* tok, const float* x_in, float* x_out, int x_on_host, int layer0,
                                                                int layer1, int head, int32_t* argmax);
/* ---- sampling.  The reference installs llama.cpp's default sampler (llama_sampling_default_params(), src/ai_models/tk_runner_lifecycle.c:76-77;
 * llama_sampling_sample, tk_runner_streaming.c:60-61; seed = tk_llm_config_t.random_seed, tk_runner_lifecycle.c:49): top-k 40, top-p 0.95,
 * min-p 0.05, temperature 0.8, one draw.  Here the default of a runner is GREEDY (the parity definition, and what the bench
 * runs); tk_mi355x_llm_runner_set_sampling switches a runner to that chain, on the device, with a counter-based generator keyed by
 * (random_seed, tokens sampled so far by this runner): the same seed gives the same ids whatever else shares the runner's passes, and the
 * oracle restates the arithmetic (orc_sample_row) so tests compare ids for fixed seeds.  temperature 0 = back to greedy.  top_k 0 = 64 (the
 * most candidates kept); top_p 1 and min_p 0 switch those filters off.  llama.cpp itself is absent (parity unpinned vs its generator). ---- */
typedef struct {
    float temperature, top_p, min_p;
    int32_t top_k;
    uint64_t seed;
    uint32_t counter; /* draws made so far with this seed */
    uint32_t reserved;
} tk_mi355x_sampling_t;
struct tk_llm_runner_s;
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_llm_runner_set_sampling(struct tk_llm_runner_s* runner, float temperature, int32_t top_k, float top_p, float min_p);
/* tk_mi355x_llm_forward with a sampling state per row (temperature <= 0: that row takes the arg max); ids [nrows] */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_llm_forward_sampled(tk_mi355x_llm_session_t* s, int nrows, const int32_t* seq, const int32_t* pos,
                                                                  const int32_t* tok, const tk_mi355x_sampling_t* sampling, float* logits, int32_t* ids);
/* KV cache import / export: positions [pos0, pos0 + n_pos) of one (layer, sequence) as IEEE f16 bits, host arrays laid out
 * [position][kv head][head_dim] (what llama.cpp's llama_state_seq_* moves for one sequence; the reference clears the cache per prompt,
 * src/ai_models/tk_runner_streaming.c:31, so it has no counterpart there).  Restores a saved prompt prefix; the parity tests use it to put
 * the attention launch at any context length in one decode step.  Synchronous. */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_llm_session_kv_write(tk_mi355x_llm_session_t* s, int layer, int seq, int pos0, int n_pos,
                                                                   const uint16_t* k, const uint16_t* v);
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_llm_session_kv_read(tk_mi355x_llm_session_t* s, int layer, int seq, int pos0, int n_pos,
                                                                  uint16_t* k, uint16_t* v);
/* equal-length prompts for sequences 0..nseq-1; first_tokens[nseq] optional */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_llm_prefill(tk_mi355x_llm_session_t* s, int nseq, int n_prompt, const int32_t* tokens,
                                                          int32_t* first_tokens);
/* greedy decode of n_steps tokens for rows 0..nrows-1, hipGraph replay; out_tokens[n_steps][tk_mi355x_llm_max_rows()] */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_llm_decode(tk_mi355x_llm_session_t* s, int nrows, int n_steps, int32_t* out_tokens,
                                                         float* ms_per_step);
/* ---- the LLM layer-sharded over GPUs, stage hand-off inside the library (BASELINE configs[4]; what is sharded is the reference's
 * llama_decode call, src/ai_models/tk_runner_streaming.c:34,77).  Stage `stage` of `n_stages` runs layers [layer0, layer1) of every pass on its
 * session's GPU; the [rows, d_model] residual stream is stored by the producer's last kernel straight into the consumer's device mailbox (mapped
 * with hipIpc across processes — one process per GPU — or by pointer inside one process: peer memory over xGMI), the last stage samples and
 * returns the ids to stage 0 the same way.  No host synchronisation, host copy or collective per pass; a stage's decode step is one captured
 * hipGraph.  payload_f16 = 0: the stream crosses as exact fp32 (tokens and logits bit-identical to one GPU); 1: IEEE f16, half the bytes, the
 * stream rounded once per boundary.  Every device-side wait is bounded (20 s; $TK_MI355X_PIPE_TIMEOUT_S seconds when set at create time):
 * after a timeout every later wait of the stage returns at once, tk_mi355x_pipe_sync returns TK_ERROR_TIMEOUT and the pipe stays failed —
 * pass / decode / sync keep failing until it is destroyed and re-created (its sequence numbers no longer agree with its neighbours').
 * The mailbox is fine-grained device memory (peer writes while kernels poll; $TK_MI355X_PIPE_COARSE=1: plain hipMalloc, one-device A/B only).
 * Any number of generations may run on one set of pipes: a pass with host-given tokens first discards the ids the previous generation left in
 * the id mailbox; a decode() needs a sampling pass (head != 0) before it.
 * All stages must enqueue the same passes in the same order.  csrc/llm/tk_llm_pipe.h has the protocol and the coherence argument. ---- */
typedef struct tk_mi355x_pipe_s tk_mi355x_pipe_t;
typedef struct { uint8_t ipc[64]; uint64_t bytes; int32_t device; int32_t pid; } tk_mi355x_pipe_handle_t; /* plain bytes: move them between processes any way */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_pipe_create(tk_mi355x_pipe_t** out, tk_mi355x_llm_session_t* s, int stage, int n_stages, int layer0, int layer1,
                                                          int payload_f16, tk_mi355x_pipe_handle_t* my_handle);
TK_API void tk_mi355x_pipe_destroy(tk_mi355x_pipe_t** p);
/* the mailboxes of stage (stage + 1) % n and (stage - 1 + n) % n, from the processes that own them */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_pipe_connect(tk_mi355x_pipe_t* p, const tk_mi355x_pipe_handle_t* next, const tk_mi355x_pipe_handle_t* prev);
/* the same for stages created in this process (several GPUs driven by one process, or the tests' two stages on one GPU) */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_pipe_connect_local(tk_mi355x_pipe_t* p, tk_mi355x_pipe_t* next, tk_mi355x_pipe_t* prev);
/* The collective form of the same hand-off (ncclSend / ncclRecv; north_star: "RCCL over xGMI only for the LLM shard"): the stages
 * form one RCCL communicator — tk_mi355x_pipe_rccl_unique_id in ONE process (128 plain bytes, moved to the others any way), then
 * tk_mi355x_pipe_connect_rccl in every stage's process INSTEAD of tk_mi355x_pipe_connect — and every boundary is an ncclSend of the exact fp32
 * stream on the producer's stream matched by an ncclRecv on the consumer's; ids return to stage 0 the same way.  One GPU per stage: with fewer
 * than two visible devices, or two stages on one device, the call FAILS (TK_ERROR_GPU_DEVICE_NOT_FOUND) — it never falls back to the
 * mailboxes.  Decode steps are launched eagerly.  pass / decode / sync are used as with the mailbox transport and give the same tokens. */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_pipe_rccl_unique_id(uint8_t out[128]);
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_pipe_connect_rccl(tk_mi355x_pipe_t* p, const uint8_t unique_id[128]);
/* enqueue one pass (returns at once).  tok: stage 0 only; NULL = feed the ids the last stage sampled for these rows.  head != 0: the last stage
 * samples and returns the ids to stage 0; every stage's device-side positions then stand one past the rows' (a decode loop may follow). */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_pipe_pass(tk_mi355x_pipe_t* p, int nrows, const int32_t* seq, const int32_t* pos, const int32_t* tok, int head);
/* enqueue n_steps greedy decode steps for the rows of the last sampling pass: n_steps replays of this stage's captured pass */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_pipe_decode(tk_mi355x_pipe_t* p, int nrows, int n_steps);
/* wait for everything enqueued.  out_tokens [n_steps][tk_mi355x_llm_max_rows()] (optional): on the last stage the ids sampled at each decode step,
 * on stage 0 the ids fed at each step (the first one = the token the prompt's sampling pass produced) */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_pipe_sync(tk_mi355x_pipe_t* p, int32_t* out_tokens, int n_steps);

/* HIP-event timing of one GEMV launch on the session stream: which = 0 gate+up, 1 down, 2 qkv, 3 lm_head, 4 o */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_llm_time_gemv(tk_mi355x_llm_session_t* s, int layer, int which, int nrows, int iters, float* avg_ms,
                                                            double* algorithmic_bytes);

/* HIP-event timing of the decode attention launch (k_attention) at nrows rows whose sequences hold ctx cached positions; kv_bytes =
 * the K / V bytes one launch must read once */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_llm_time_attention(tk_mi355x_llm_session_t* s, int nrows, int ctx, int iters, float* avg_ms,
                                                                 double* kv_bytes);

/* which attention launch a pass of `nrows` rows takes on `device` (the choice depends on its CU count): out = {kernel, query heads per
 * workgroup, positions per ring slot / resident chunk, ring slots}; kernel 0 = k_attention (ring of chunks), 1 = k_attention_narrow (<= one
 * workgroup per CU, the context resident in LDS), 2 = k_attention_prefill (fused == 0 only: 16 rows of a sequence per workgroup on the fp32
 * matrix pipe, taken by passes that reach position 128 or beyond; the other three numbers describe the k_attention form that shorter contexts
 * take and that TK_MI355X_NO_PREFILL_ATT=1 puts back everywhere — the two forms are bit-identical).  fused != 0: a decode
 * pass (every sequence once), else a pass that holds several positions of a sequence (prompt chunks).  The parity
 * tests assert the instantiation they exercise from this answer (csrc/llm/tk_llm_kernels.hip: tk_attention_plan). */
TK_API int tk_mi355x_device_cu_count(int device); /* compute units of a HIP device (256 on an MI355X), -1 when there is no such device */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_attention_plan(int device, int nrows, int n_head, int n_kv_head, int head_dim, int max_ctx, int fused, int32_t out[4]);
/* the same for a pass whose highest position is `top_position` (a session picks per pass from the positions it is handed): kernel 3 = the
 * long-context decode form (fused != 0; 1 .. 4 rows from position 512, 5 .. 8 rows from 768: k_att_scores_long +
 * k_att_pv_chain + k_att_pv_join — scores over (row, KV head, 64-position block) workgroups, one PV chain per (row, head, class) wave — instead
 * of one latency chain per pair of heads; $TK_MI355X_NO_LONG_ATT=1 keeps the fused kernels), kernel 2 only from position 128 on.  All forms
 * are bit-identical. */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_attention_plan_at(int device, int nrows, int n_head, int n_kv_head, int head_dim, int max_ctx, int fused,
                                                                int top_position, int32_t out[4]);

/* ---- tool-call grammar (GBNF) — the sampling constraint of tk_llm_runner_prepare_generation(..., use_tool_grammar = true)
 * (reference: src/ai_models/grammars/tool_call.gbnf via llama.cpp's grammar sampler, tk_runner_lifecycle.c:59,
 * tk_runner_streaming.c:44-48,69-75).  gbnf == NULL selects the built-in tool-call grammar.  No GPU involved. ---- */
/* how many leading bytes of `text` the grammar accepts, and whether the grammar is complete after them */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_grammar_check(const char* gbnf, const char* text, int32_t* n_accepted, int32_t* complete);
/* allowed[b] = 1 when byte b may follow `prefix` (which must itself be accepted) */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_grammar_next_bytes(const char* gbnf, const char* prefix, uint8_t allowed[256], int32_t* complete);
/* text accepted by the grammar in the current generation: the tool call to read after tk_llm_runner_generate_next_token
 * returned the sentinel (const char*)1; valid until the next prepare_generation */
struct tk_llm_runner_s;
TK_API const char* tk_mi355x_llm_runner_tool_call_text(struct tk_llm_runner_s* runner);

/* test hook for the two exact fp32 GEMMs: C[M][N] = act(A[M][K] W[N][K]^T + bias) + residual on host buffers, once through the LDS-staged
 * kernel (k_gemm_f32 / k_gemm_f32_big) into c_staged and once through the tiled kernel (csrc/nn/tk_gemm_tiled.h: weights as f32 tiles, or
 * as f16 tiles with f16-rounded activations when f16 != 0) into c_tiled.  K must be a multiple of 128.  bias / residual may be NULL;
 * act: 0 none, 1 SiLU, 2 GELU, 3 sigmoid. */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_gemm_pair(int device, int M, int N, int K, const float* a, const float* w, const float* bias,
                                                        const float* residual, int act, int f16, float* c_staged, float* c_tiled);

#ifdef __cplusplus
}
#endif
#endif
