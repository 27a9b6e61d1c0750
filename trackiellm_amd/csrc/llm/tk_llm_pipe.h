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
 * Every wait is bounded (TK_PIPE_TIMEOUT_S seconds of s_memrealtime, or $TK_MI355X_PIPE_TIMEOUT_S): a peer that never publishes sets
 * the pipe's status word; from then on every wait of this stage returns at once and no counter, slot or credit moves, so whatever is
 * still enqueued drains in microseconds, the host sees TK_ERROR_TIMEOUT at its next sync and the pipe stays failed (re-create it).
 *
 * Coherence (who caches what; ADVICE r03 / VERDICT r03 item 4).  The protocol needs visibility BETWEEN AGENTS IN THE MIDDLE OF KERNELS:
 * a peer GPU stores into this GPU's mailbox over xGMI while a kernel resident here polls a flag word of it.
 *  - Memory type.  HIP guarantees coherence of plain hipMalloc ("coarse-grained") memory between agents only at kernel boundaries: its
 *    pages are mapped so that the HOME GPU's L2s may keep lines of it on the assumption that nobody else writes them while a kernel
 *    runs, and a system-scope acquire (buffer_inv sc0 sc1) is not required to drop such lines.  A peer's write lands in HBM behind those
 *    L2s: a poller could spin on a stale flag for ever, a reader could get a slot's contents of eight passes ago.  On ONE device (every test
 *    of this build, profiles/r03_multirank_rehearsal.txt) all stages sit behind the same L2s and HBM, which hides that.  So the mailbox —
 *    flags, credits, id payload AND the x payload slots (one allocation, one IPC handle) — is FINE-GRAINED device memory
 *    (hipExtMallocWithFlags(hipDeviceMallocFinegrained)), the type HIP documents as coherent between agents at system-scope
 *    synchronisation points inside a running kernel: it is mapped so that neither the home GPU's nor a peer's L2 holds a line of it
 *    past a system-scope release / acquire.  The per-stage private state (TkPipeState) is only ever touched by its own GPU and stays
 *    coarse-grained.
 *  - Producer (k_pipe_send_x / k_pipe_send_ids, runs on the PEER of the mailbox's home): plain payload stores into the mapped slot ->
 *    __threadfence_system() by every storing thread (release at system scope: write-back of this GPU's L2 lines of remote fine-grained
 *    memory, s_waitcnt vmcnt(0)) -> workgroup barrier -> arrival counter in the producer's own memory -> the LAST workgroup fences again
 *    and stores the slot's sequence number with a system-scope release store.  The flag can therefore not overtake any workgroup's
 *    payload: each fenced its stores before it arrived.
 *  - Consumer (k_pipe_wait_x / k_pipe_take_x / k_pipe_take_ids, on the home GPU): ONE lane polls the flag with system-scope acquire loads
 *    (they bypass the L1 and, on fine-grained memory, the L2), then EVERY workgroup that reads payload executes a system-scope acquire
 *    fence before its first payload load (it invalidates that CU's L1 and the non-coherent lines of its XCD's L2) — the poller's acquire
 *    covers only the poller's own CU and XCD, and a launch boundary does not invalidate peer-written lines.  This is the fix of commit
 *    919aa78 (stale slot contents behind other XCDs' L2s were seen ON ONE DEVICE with coarse-grained memory); with fine-grained memory
 *    the fence is still required for the L1.
 *  - Credits flow the other way under the same two rules (system-scope release store by the consumer, acquire poll by the producer).
 * Status: the fine-grained mailbox passes the one-process (pointer) and two-process (hipIpc) tests on one GPU; NO run has crossed a
 * device boundary yet (the GPU boxes of this build have one GPU) — multi-GPU numbers and claims stay unverified until
 * tests/test_pipeline_gpu.py::test_in_library_handoff_two_processes_ipc_mapped_mailboxes has passed once with one stage per device.
 */
#ifndef TK_LLM_PIPE_H
#define TK_LLM_PIPE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <deque>
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
    int status;            /* 0 ok, 1 a wait timed out: sticky — every later wait returns at once, no counter moves any more */
    int pad;
    unsigned long long timeout_ticks; /* bound of one wait in s_memrealtime ticks (100 MHz): TK_PIPE_TIMEOUT_S, or $TK_MI355X_PIPE_TIMEOUT_S */
};

struct TkPipeHandle { /* what the host exchanges between stages: 64 handle bytes + the block size + who made it */
    unsigned char ipc[64];
    uint64_t bytes;
    int32_t device;
    int32_t pid;
};

/* 128 bytes that identify a new RCCL communicator (ncclGetUniqueId); made by ONE process, handed to every stage.  false + *err when librccl
 * cannot be loaded. */
bool tk_pipe_rccl_unique_id(unsigned char out[128], std::string* err);

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
    /* the collective form of the hand-off (ncclSend / ncclRecv is the path's primary collective, north_star: "RCCL over xGMI only
     * for the LLM shard"): instead of mapping mailboxes, the n_stages ranks form one RCCL communicator (unique id from
     * tk_pipe_rccl_unique_id of stage 0's process, any channel) and every boundary is an ncclSend of the folded fp32 stream on the producer's
     * stream matched by an ncclRecv into the consumer's residual buffer; the sampled ids return to stage 0 the same way.  One GPU per stage:
     * refused when fewer than two devices are visible or when two stages sit on one device (RCCL rejects duplicate devices).  fp32 payload
     * only; decode steps are launched eagerly (no graph replay: the insurance path, and the cross-check of the mailbox path on real xGMI). */
    bool connect_rccl(const unsigned char unique_id[128]);
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
    bool fine_grained_ = true;         /* mailbox in fine-grained device memory (see "Coherence" above) */
    bool failed_ = false;              /* a wait timed out: sticky until the pipe is destroyed */
    int ids_outstanding_ = 0;          /* id messages the last stage has been asked to send minus those stage 0 has been asked to take */
    int32_t* h_rows_ = nullptr;        /* pinned staging ring for (seq, pos, tok) of host-described passes */
    int h_next_ = 0;
    hipGraphExec_t graph_[2][TK_MAX_ROWS + 1] = {}; /* [long-context attention form][row count] */
    int host_top_ = 0;                 /* highest position of the rows decode() continues (host-side count; the device advances its own) */
    void* rccl_comm_ = nullptr;        /* ncclComm_t when the RCCL transport is selected */
    int32_t* rccl_scratch_ = nullptr;  /* [TK_MAX_ROWS] ids of a drained message */
    std::deque<int> id_msg_rows_;      /* RCCL transport, stage 0: row counts of the id messages not yet received (sends and receives must pair) */
    void enqueue_stage(int nrows, bool take_ids, bool head, bool advance_pos, bool fused_attn);
    uint8_t* x_payload(TkPipeBlock* b) const { return (uint8_t*)b + sizeof(TkPipeBlock); }
};

#endif
