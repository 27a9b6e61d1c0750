/*
 * tk_llm_engine.h — device-resident Mistral-class model + batched decode session.
 *
 * Replaces what the reference obtains from llama.cpp:
 *   model  <-> llama_load_model_from_file            (src/ai_models/tk_model_loader.c:245-251)
 *   session<-> llama_new_context_with_model + KV     (src/ai_models/tk_runner_lifecycle.c:47-51)
 *   forward<-> llama_decode(batch)                   (src/ai_models/tk_runner_streaming.c:34,77)
 * One model is shared by any number of sessions; a session owns a KV cache for `max_seq`
 * sequences and runs up to 16 (sequence, position) rows per pass so that B concurrent
 * cortex cycles read the 4.3 GB of weights once per step.
 */
#ifndef TK_LLM_ENGINE_H
#define TK_LLM_ENGINE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "tk_llm_kernels.h"

struct TkLlmHParams {
    int32_t n_layer, d_model, n_head, n_kv_head, head_dim, d_ff, vocab;
    float rms_eps, rope_theta;
    /* K-split plan: part of the canonical summation order shared with the oracle */
    int32_t ks_qkv, ks_o, ks_gateup, ks_down, ks_out;
};

struct TkDevTensor {
    uint8_t* data = nullptr; /* tiles (matrices) or raw f32 / GGUF blocks */
    int type = 0;
    int64_t rows = 0, cols = 0;
    size_t bytes = 0;
};

struct TkLlmLayer {
    TkDevTensor attn_norm, ffn_norm, q, k, v, o, gate, up, down;
};

enum TkLlmTensorId { TK_T_TOKEN_EMBD = 0, TK_T_OUT_NORM = 1, TK_T_OUTPUT = 2 };
enum TkLlmLayerTensorId { TK_L_ATTN_NORM = 0, TK_L_Q, TK_L_K, TK_L_V, TK_L_O, TK_L_FFN_NORM, TK_L_GATE, TK_L_UP, TK_L_DOWN, TK_L_COUNT };

class TkLlmModel {
public:
    TkLlmHParams hp{};
    int device = 0;
    TkDevTensor token_embd, out_norm, output;
    std::vector<TkLlmLayer> layers;
    std::string error;
    size_t weight_bytes = 0; /* bytes a decode step streams (matrices only) */

    ~TkLlmModel();
    /* returns false and sets `error` on failure */
    bool init(const TkLlmHParams& hp, int device);
    /* f16: the fp16 checkpoint recipe (every matrix and the embedding IEEE f16, norms f32) instead of Q4_K_M */
    bool fill_synthetic(uint64_t seed, bool f16 = false);
    bool has_f16 = false; /* some matrix is f16: sessions also keep f16-rounded f32 activations */
    /* a LoRA adapter to merge into every matrix it names WHILE that matrix is installed (set before set_tensor / fill_synthetic; tk_lora.h);
     * not owned, only read during those calls */
    const struct TkLoraAdapter* lora = nullptr;
    int lora_merged = 0; /* matrices the adapter changed */
    /* `host_blocks` is the tensor in GGUF layout (F32 for norms) */
    bool set_tensor(int layer, int which, int type, const void* host_blocks, size_t nbytes);
    bool ready() const;
    static int recipe_type(const TkLlmHParams& hp, int layer, int which);
    void shape(int layer, int which, int64_t* rows, int64_t* cols) const;

private:
    TkDevTensor* slot(int layer, int which);
    bool install(TkDevTensor* t, int type, int64_t rows, int64_t cols, void* dev_blocks, hipStream_t s, int layer, int which);
};

class TkLlmPipe;

class TkLlmSession {
public:
    TkLlmModel* model = nullptr;
    int max_seq = 0, max_ctx = 0;
    std::string error;
    hipStream_t stream = nullptr;

    ~TkLlmSession();
    bool init(TkLlmModel* m, int max_seq, int max_ctx);
    /* one pass over nrows <= TK_MAX_ROWS rows (16-row M-tiles); host arrays.  row_masks (optional): row_masks[r] = nullptr or the
     * (vocab + 31) / 32 words whose bit t says token t may be sampled for row r: grammar-constrained greedy sampling, per row, so
     * constrained and unconstrained rows share passes (and the captured graphs).  row_samp (optional): [nrows] sampling state, temp > 0
     * = the reference's default stochastic chain for that row (tk_llm_kernels.h), else greedy */
    bool forward(int nrows, const int32_t* seq, const int32_t* pos, const int32_t* tok, float* logits_host, int32_t* argmax_host,
                 bool lm_head = true, const uint32_t* const* row_masks = nullptr, const TkSampleRow* row_samp = nullptr);
    /* one pipeline stage of a pass: layers [l0, l1) on this GPU.  The first stage starts from `tok` (x_in == nullptr), later stages
     * from the residual stream x_in [nrows][d_model] fp32; every stage but the last writes the stream to x_out; the last one
     * (head == true, l1 == n_layer) samples.  x_on_host: x_in / x_out are host pointers (gloo transport), otherwise device
     * pointers (RCCL transport).  Bit-identical to forward() whatever the split. */
    bool forward_stage(int nrows, const int32_t* seq, const int32_t* pos, const int32_t* tok, const float* x_in, float* x_out, bool x_on_host,
                       int l0, int l1, bool head, int32_t* argmax_host);
    /* prompts of equal length for sequences 0..nseq-1 (tokens[nseq][n_prompt]); leaves row r = sequence r
     * holding the first sampled token so decode() can follow; first_tokens_host[nseq] optional */
    bool prefill(int nseq, int n_prompt, const int32_t* tokens, int32_t* first_tokens_host);
    /* greedy decode loop, on-device feedback, hipGraph replay.  Starts from the rows' current
     * (tok, pos) state left by the previous forward()/decode(); out_tokens[n_steps][nrows]. */
    bool decode(int nrows, int n_steps, int32_t* out_tokens_host);
    bool reset();
    /* KV cache import / export: rows [pos0, pos0 + n_pos) of one (layer, sequence) as f16 bits, host arrays in [position][kv head][dim]
     * order (the cache itself is [layer][sequence][kv head][position][dim]).  Restores a saved prompt prefix without re-running it; the
     * parity tests use it to put an attention launch at any context length in one decode step. */
    bool kv_write(int layer, int seq, int pos0, int n_pos, const uint16_t* k, const uint16_t* v);
    bool kv_read(int layer, int seq, int pos0, int n_pos, uint16_t* k, uint16_t* v);
    /* per-launch timing of the last decode(): average ms of one step measured with HIP events */
    float last_step_ms = 0.0f;
    /* stand-alone timing of the dominant GEMV (gate/up of layer 0) for bench.py's roofline leg */
    bool time_gemv(int layer, int which, int nrows, int iters, float* avg_ms, double* algo_bytes);
    /* stand-alone timing of the decode attention launch at nrows rows x ctx cached positions, for bench.py's second roofline object */
    bool time_attention(int nrows, int ctx, int iters, float* avg_ms, double* kv_bytes);

private:
    friend class TkLlmPipe; /* tk_llm_pipe.h: a pipeline stage enqueues layer ranges of this session between its hand-off kernels */
    int last_ks_res = 1;    /* slabs of the residual update the last enqueue_range left pending in `partial` */
    void enqueue_pass(int nrows, bool lm_head, bool fused_attn);
    void enqueue_range(int nrows, int l0, int l1, bool embed, bool fold_out, bool lm_head, bool fused_attn);
    int enqueue_matmul(const TkDevTensor* const* t, int nseg, int K, int ks, int n_total, const TkActQ8& act, float* out, int nrows);
    uint16_t *kcache = nullptr, *vcache = nullptr;
    float *x = nullptr, *x2 = nullptr, *qbuf = nullptr, *partial = nullptr, *partial2 = nullptr, *logits = nullptr, *rope_cos = nullptr, *rope_sin = nullptr;
    TkActQ8 act_d{}, act_qd{}, act_ff{};
    int32_t *d_seq = nullptr, *d_pos = nullptr, *d_tok = nullptr, *d_nsteps = nullptr, *d_hist = nullptr;
    uint32_t* d_mask = nullptr;     /* [TK_MAX_ROWS][(vocab + 31) / 32] allowed-token bits of the masked rows of the current pass */
    int32_t* d_mask_row = nullptr;  /* [TK_MAX_ROWS] index into d_mask, -1 = the row samples unconstrained */
    bool mask_rows_dirty = true;    /* d_mask_row holds something other than all -1 */
    TkSampleRow* d_samp = nullptr;  /* [TK_MAX_ROWS] sampling state of the rows of the current pass (temp 0 = greedy); decode() continues from it */
    bool samp_dirty = false;        /* d_samp holds a stochastic row */
    std::string launch_error; /* set by enqueue_* when a launcher refuses its arguments (no HIP error is raised for that) */
    int hist_cap = 0;
    hipGraphExec_t graph_exec[2][TK_MAX_ROWS + 1] = {}; /* [long_pass][row count]: decode pass (head + sampling, every sequence once) */
    /* [tiled_pass][row count]: prompt pass (no head, rope/append as its own kernel); sampling pass that holds several positions of one sequence
     * (a prompt's last chunk) */
    hipGraphExec_t graph_prefill[2][TK_MAX_ROWS + 1] = {};
    hipGraphExec_t graph_head_nf[2][TK_MAX_ROWS + 1] = {};
    /* set by whoever describes a pass, read by enqueue_range: a multi-position pass that reaches position TK_TILED_ATT_MIN_POS or beyond takes
     * k_attention_prefill (16 rows of a sequence per workgroup), shorter contexts k_attention's per-row form (3 us per launch quicker below ~128
     * positions, profiles/r05_prefill_attention.txt); the two are bit-identical, so the choice never shows in a result */
    bool tiled_pass = false;
    /* a decode pass (every sequence once) of at most TK_LONG_ATT_MAX_ROWS rows that reaches position TK_LONG_ATT_MIN_POS runs its attention as
     * append + scores + PV launches spread over the chip (tk_launch_attention_long) instead of one latency chain per pair of heads; bit-identical */
    bool long_pass = false;
    bool counted_ = false; /* this session is in the device's count of live decode sessions (tk_attention_note_session) */
    float* d_scores = nullptr; /* [TK_LONG_ATT_MAX_ROWS][n_head][max_ctx], allocated when the window can reach TK_LONG_ATT_MIN_POS */
    void choose_attention(const int32_t* pos, int nrows);
    void choose_attention_top(int top, int nrows);
    int32_t* d_tab = nullptr;                           /* prefill schedule: [3][total rows] = seq, pos, tok */
    int32_t* d_tiles = nullptr;                         /* [1 + TK_MAX_ROWS] 16-row tiles of a multi-position pass (k_att_tiles) */
    size_t tab_cap = 0;
    bool capture_pass(hipGraphExec_t* slot, int nrows, bool lm_head, bool fused_attn);
public:
    uint64_t n_captures = 0; /* passes recorded into a graph so far, and the host time that took (diagnostics: TK_MI355X_BATCHER_TRACE) */
    double capture_ms = 0.0;
private:
};

#endif
