/*
 * tk_llm_batcher.h — continuous batching behind the reference's one-sequence-per-runner API.
 *
 * The reference gives every tk_llm_runner_t its own llama context and decodes one token per call on the caller's thread
 * (src/ai_models/tk_runner_streaming.c:57-85, contract src/ai_models/tk_model_runner.h:164-181).  Behind the same calls, all runners
 * created on one model share ONE decode session here: a runner owns a sequence slot of the session's KV cache; prepare_generation /
 * generate_next_token / add_tool_response enqueue their (sequence, position, token) rows and block; one scheduler thread per model
 * coalesces whatever is ready — prompt chunks and single decode rows of different runners alike — into passes of up to 256 rows, so
 * K host threads that each drive "their" runner read the 4.3 GB of weights once per step instead of K times.
 * A row's result does not depend on which other rows share its pass (bit-identical logits at any pass width:
 * tests/test_llm_gpu.py::test_full_7b_pass_width_invariance), so a runner sees exactly the tokens it would see alone.
 *
 * Run-ahead.  generate_next_token is a host round trip per token: the owner wakes, turns the id into a piece, returns to its caller and
 * comes back with the id it was just given — while the GPU waits.  So when a sequence's sampled id has been handed to its owner, the
 * scheduler feeds that id at the next position in its very next pass WITHOUT waiting for the owner (one position ahead, never more); the
 * owner's next call then finds its row in flight or already done.  Not for rows sampled under a grammar mask (the next mask depends on
 * the accepted token), not after EOS, not at the end of the context.  An owner that comes back with something else (a tool response, a
 * new prompt) or not at all costs one wasted row: the cache row it wrote is overwritten before anything attends to it, because positions
 * are fed in order.  Tokens are unchanged — a row's result does not depend on when or with whom it runs.
 */
#ifndef TK_LLM_BATCHER_H
#define TK_LLM_BATCHER_H

#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "tk_llm_engine.h"

class TkLlmBatcher {
public:
    ~TkLlmBatcher();
    /* one session with `slots` sequences of `n_ctx` positions on the model's device; starts the scheduler thread */
    bool init(TkLlmModel* model, int slots, int n_ctx, int32_t eos, std::string* err);
    int acquire_slot();            /* -1 when every slot is taken */
    void release_slot(int slot);
    int n_ctx() const { return n_ctx_; }
    int slots() const { return (int)slot_used_.size(); }
    /* Blocking: feed `n` tokens of sequence `slot` at positions pos0, pos0 + 1, ...; *sampled = arg max after the last one (over the
     * tokens `mask` allows, when given: (vocab + 31) / 32 words, bit t = token t).  Thread-safe; one outstanding call per slot. */
    /* samp (optional): sampling state of the sampled row — temp > 0 draws from the reference's default stochastic chain with the given
     * (seed, counter) instead of the arg max; such rows never run ahead (the next counter is the owner's) */
    bool submit(int slot, int pos0, const int32_t* toks, int n, const uint32_t* mask, int32_t* sampled, std::string* err, const TkSampleRow* samp = nullptr);
    /* counters for tests and bench: passes run, rows processed FOR AN OWNER (a run-ahead row counts when its owner takes it; wasted ones
     * are in *wasted), the widest pass so far */
    void stats(uint64_t* passes, uint64_t* rows, int* max_rows, uint64_t* wasted = nullptr);

private:
    struct Request {
        int slot, pos0, n, done_rows = 0;
        const int32_t* toks;
        const uint32_t* mask;
        TkSampleRow samp{};
        int32_t sampled = -1;
        bool finished = false, ok = true;
        std::string error;
        std::condition_variable cv;
    };
    /* one run-ahead row per sequence slot */
    struct Ahead {
        enum State { NONE, PLANNED, INFLIGHT, DONE } st = NONE;
        int pos = 0;
        int32_t tok = 0, sampled = -1;
        bool ok = true, discard = false;
        Request* waiter = nullptr;
    };
    void loop();
    void plan_ahead(int slot, int pos_done, int32_t sampled, bool masked); /* mu_ held */
    void drop_ahead(int slot);                                              /* mu_ held */
    std::vector<Ahead> ahead_;
    int32_t eos_ = -1;
    uint64_t wasted_ = 0;
    std::string last_error_;
    TkLlmSession session_;
    int n_ctx_ = 0;
    std::vector<char> slot_used_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Request*> queue_;
    bool stop_ = false;
    bool decode_first_ = false; /* TK_MI355X_BATCHER_DECODE_FIRST=1: sampled rows before prompt rows when a pass is formed (A/B; tk_llm_batcher.cpp) */
    size_t expect_ = 0; /* requests finished by the last pass: their owners are about to submit the next token */
    uint64_t passes_ = 0, rows_ = 0;
    int max_rows_ = 0;
    /* TK_MI355X_BATCHER_TRACE=<file>: one line per pass (times in ms since the scheduler started: woken, pass formed, pass done; rows, of them
     * run-ahead rows, requests completed, requests still queued, graph captures so far and their host time), written when the batcher dies */
    struct PassTrace { double woke, formed, done; int rows, ahead, completing, queued_after; uint64_t captures; double capture_ms; };
    std::vector<PassTrace> trace_;
    std::string trace_path_;
    std::thread worker_;
};

#endif
