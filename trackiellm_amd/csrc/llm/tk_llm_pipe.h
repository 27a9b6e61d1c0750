/*
 * tk_llm_pipe.h — the LLM layer-sharded over GPUs with the stage hand-off INSIDE the library (BASELINE configs[4]).
 *
 * What is sharded is the reference's `llama_decode` call (src/ai_models/tk_runner_streaming.c:34,77): stage s of n runs layers
 * [l0, l1) of every pass on its own GPU (one process per GPU, or several stages in one process), the [rows, d_model] residual stream moves
 * to stage s + 1, the last stage samples and returns the token ids to stage 0.  No all-reduce exists on this path.
 *
 * Transport: every stage owns one device block ("mailbox") that its two ring neighbours map — across processes with
 * hipIpcGetMemHandle / hipIpcOpenMemHandle (dmabuf IPC; a peer GPU's memory then is reached over the direct xGMI link), inside one
 * process by pointer.  A producer's LAST KERNEL of a pass (k_pipe_send_x: it also finishes the pending residual update, what
 * k_residual_fold does) stores the stream straight into a slot of the consumer's mailbox, fences at system scope and publishes the slot's
 * sequence number; the consumer's FIRST KERNEL of the pass (k_pipe_wait) polls that number on its own stream.  Slots are recycled under
 * credits the consumer writes back into the producer's block, so a stage may run several passes ahead (prompt chunks) without
 * overwriting anything.  Sequence numbers live in device memory and advance inside the kernels: a stage's pass is a captured
 * hipGraph that is replayed once per decode step with NO host synchronisation, host copy or collective call per pass.
 * Payload: exact fp32 (default: the pipeline's tokens and logits are bit-identical to one GPU) or IEEE f16 (SURVEY §8e's 8 KiB per
 * row; the residual stream is rounded once per boundary — a tolerance mode).
 * Every wait is bounded (TK_PIPE_TIMEOUT_S seconds of s_memrealtime): a peer that never publishes sets the pipe's status word and
 * the pass drains; the host sees the error at its next synchronisation, nothing hangs.
 */
#ifndef TK_LLM_PIPE_H
#define TK_LLM_PIPE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "tk_llm_engine.h"

#define TK_PIPE_SLOTS 8
#define TK_PIPE_TIMEOUT_S 20

/* the device block a stage exports.  Who writes what:
 *   x_flag / x_payload     stage s - 1 (its k_pipe_send_x), read by s
 *   ids_flag / ids_payload the last stage (k_pipe_send_ids) into stage 0's block, read by stage 0
 *   credit_x               stage s + 1: how many of s's x messages it has consumed
 *   credit_ids             stage 0 into the last stage's block: how many id messages it has consumed */
struct TkPipeBlock {
    unsigned long long x_flag[TK_PIPE_SLOTS][16];   /* one 128-byte line per flag */
    unsigned long long ids_flag[TK_PIPE_SLOTS][16];
    unsigned long long credit_x[16];
    unsigned long long credit_ids[16];
    int32_t ids_payload[TK_PIPE_SLOTS][TK_MAX_ROWS];
    /* x payload follows: TK_PIPE_SLOTS x TK_MAX_ROWS x d_model x 4 bytes */
};

/* per-stage private device state: counters that advance inside the kernels (graph replays carry no sequence numbers) */
struct TkPipeState {
    unsigned long long sent_x, recv_x, sent_ids, recv_ids;
    unsigned int done_x;   /* arrival counter of k_pipe_send_x's workgroups */
    unsigned int done_take;
    int status;            /* 0 ok, 1 a wait timed out */
    int pad;
};

struct TkPipeHandle { /* what the host exchanges between stages: 64 handle bytes + the block size + who made it */
    unsigned char ipc[64];
    uint64_t bytes;
    int32_t device;
    int32_t pid;
};

class TkLlmPipe {
public:
    std::string error;
    ~TkLlmPipe();
    /* stage `stage` of `n_stages` runs layers [l0, l1) on `session`'s GPU and stream; allocates and exports the mailbox */
    bool init(TkLlmSession* session, int stage, int n_stages, int l0, int l1, bool payload_f16, TkPipeHandle* out_handle);
    /* map the ring neighbours' mailboxes: `next` = stage (s + 1) % n, `prev` = stage (s - 1 + n) % n (handles from other processes) */
    bool connect(const TkPipeHandle* next, const TkPipeHandle* prev);
    /* the same for stages that live in this process (pointers; enables peer access when the devices differ) */
    bool connect_local(TkLlmPipe* next, TkLlmPipe* prev);
    /* enqueue one pass of this stage on the session's stream (returns without waiting for the GPU).  tok: stage 0 only — the rows' tokens,
     * or NULL to take them from the id mailbox (the ids the last stage sampled for these rows' previous positions).  head: the last stage
     * runs the lm head, samples and sends the ids to stage 0. */
    bool pass(int nrows, const int32_t* seq, const int32_t* pos, const int32_t* tok, bool head);
    /* enqueue n_steps greedy decode steps for rows 0 .. nrows - 1 (row r = sequence seq0 + r) continuing from the rows' device-side
     * positions: the captured graph of this stage's pass, replayed n_steps times; tokens flow stage 0 <- last through the id mailbox */
    bool decode(int nrows, int n_steps);
    /* wait for everything enqueued; fails when a wait kernel timed out.  out_tokens [n_steps][TK_MAX_ROWS] of the decode() since the last
     * call: on the last stage the ids sampled at each step (what TkLlmSession::decode returns), on stage 0 the ids FED at each step (the
     * first one is the token the prompt's sampling pass produced) */
    bool sync(int32_t* out_tokens, int n_steps);
    int stage = 0, n_stages = 1, l0 = 0, l1 = 0;

private:
    friend class TkLlmSession;
    TkLlmSession* s_ = nullptr;
    bool f16_ = false;
    TkPipeBlock* mine_ = nullptr;      /* this stage's mailbox (device memory on s_'s GPU) */
    size_t block_bytes_ = 0;
    TkPipeBlock *next_ = nullptr, *prev_ = nullptr; /* the neighbours' mailboxes as mapped here */
    bool next_ipc_ = false, prev_ipc_ = false;
    TkPipeState* st_ = nullptr;
    int32_t* h_rows_ = nullptr;        /* pinned staging ring for (seq, pos, tok) of host-described passes */
    int h_next_ = 0;
    hipGraphExec_t graph_[TK_MAX_ROWS + 1] = {};
    void enqueue_stage(int nrows, bool take_ids, bool head, bool advance_pos, bool fused_attn);
    uint8_t* x_payload(TkPipeBlock* b) const { return (uint8_t*)b + sizeof(TkPipeBlock); }
};

#endif
