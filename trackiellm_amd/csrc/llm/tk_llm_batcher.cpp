#include "tk_llm_batcher.h"

#include <algorithm>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>

TkLlmBatcher::~TkLlmBatcher() {
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
    }
    cv_.notify_all();
    if (worker_.joinable()) worker_.join();
    if (!trace_path_.empty()) {
        if (FILE* f = fopen(trace_path_.c_str(), "a")) {
            fprintf(f, "# batcher %p: woke_ms formed_ms done_ms rows ahead completing queued_after captures capture_ms\n", (void*)this);
            for (const PassTrace& t : trace_)
                fprintf(f, "%.3f %.3f %.3f %d %d %d %d %llu %.2f\n", t.woke, t.formed, t.done, t.rows, t.ahead, t.completing, t.queued_after, (unsigned long long)t.captures, t.capture_ms);
            fclose(f);
        }
    }
}

bool TkLlmBatcher::init(TkLlmModel* model, int slots, int n_ctx, int32_t eos, std::string* err) {
    if (slots < 1) slots = 1;
    if (slots > TK_MAX_ROWS) slots = TK_MAX_ROWS; /* one decode row per runner must fit one pass */
    if (!session_.init(model, slots, n_ctx)) { *err = session_.error; return false; }
    n_ctx_ = n_ctx;
    slot_used_.assign((size_t)slots, 0);
    ahead_.assign((size_t)slots, Ahead());
    eos_ = eos;
    if (const char* e = getenv("TK_MI355X_BATCHER_DECODE_FIRST")) decode_first_ = e[0] == '1';
    if (const char* e = getenv("TK_MI355X_BATCHER_TRACE")) { trace_path_ = e; trace_.reserve(4096); }
    worker_ = std::thread([this] { loop(); });
    return true;
}

int TkLlmBatcher::acquire_slot() {
    std::lock_guard<std::mutex> lk(mu_);
    for (size_t i = 0; i < slot_used_.size(); ++i)
        if (!slot_used_[i]) { slot_used_[i] = 1; return (int)i; }
    return -1;
}

void TkLlmBatcher::release_slot(int slot) {
    std::lock_guard<std::mutex> lk(mu_);
    if (slot >= 0 && slot < (int)slot_used_.size()) { slot_used_[(size_t)slot] = 0; drop_ahead(slot); }
}

void TkLlmBatcher::stats(uint64_t* passes, uint64_t* rows, int* max_rows, uint64_t* wasted) {
    std::lock_guard<std::mutex> lk(mu_);
    if (passes) *passes = passes_;
    if (rows) *rows = rows_;
    if (max_rows) *max_rows = max_rows_;
    if (wasted) *wasted = wasted_;
}

/* the owner of `slot` has just been handed `sampled` for position pos_done: feed it at pos_done + 1 in the next pass */
void TkLlmBatcher::plan_ahead(int slot, int pos_done, int32_t sampled, bool masked) {
    Ahead& a = ahead_[(size_t)slot];
    if (masked || sampled < 0 || sampled == eos_ || pos_done + 2 >= n_ctx_ || a.st != Ahead::NONE) return; /* the runner stops at n_past + 1 >= n_ctx */
    a = Ahead();
    a.st = Ahead::PLANNED;
    a.pos = pos_done + 1;
    a.tok = sampled;
}

void TkLlmBatcher::drop_ahead(int slot) {
    Ahead& a = ahead_[(size_t)slot];
    if (a.st == Ahead::NONE) return;
    if (a.st == Ahead::INFLIGHT) { a.discard = true; a.waiter = nullptr; return; } /* the pass that carries it retires it */
    if (a.st == Ahead::DONE) wasted_++;
    a = Ahead();
}

bool TkLlmBatcher::submit(int slot, int pos0, const int32_t* toks, int n, const uint32_t* mask, int32_t* sampled, std::string* err, const TkSampleRow* samp) {
    if (n <= 0) { *err = "nothing to feed"; return false; }
    if (slot < 0 || slot >= (int)slot_used_.size() || pos0 < 0 || pos0 + n > n_ctx_) { *err = "rows do not fit the context window"; return false; }
    Request r;
    r.slot = slot; r.pos0 = pos0; r.n = n; r.toks = toks; r.mask = mask;
    if (samp) r.samp = *samp;
    const bool stochastic = r.samp.temp > 0.0f;
    std::unique_lock<std::mutex> lk(mu_);
    if (stop_) { *err = "scheduler stopped"; return false; }
    Ahead& a = ahead_[(size_t)slot];
    if (a.st != Ahead::NONE && !a.discard) {
        if (n == 1 && !mask && !stochastic && pos0 == a.pos && toks[0] == a.tok) { /* the row this call asks for is the one that ran (or runs) ahead */
            if (a.st == Ahead::DONE) {
                const bool ok = a.ok;
                const int32_t got = a.sampled;
                a = Ahead();
                if (!ok) { *err = last_error_; return false; }
                rows_++;
                *sampled = got;
                plan_ahead(slot, pos0, got, false);
                cv_.notify_all();
                return true;
            }
            a.waiter = &r; /* PLANNED or INFLIGHT: the pass that carries the row finishes this request */
            cv_.notify_all();
            r.cv.wait(lk, [&] { return r.finished; });
            if (!r.ok) { *err = r.error; return false; }
            *sampled = r.sampled;
            return true;
        }
        drop_ahead(slot); /* the owner went another way (a tool response, a new prompt, a masked step) */
    }
    queue_.push_back(&r);
    cv_.notify_all();
    r.cv.wait(lk, [&] { return r.finished; });
    if (!r.ok) { *err = r.error; return false; }
    *sampled = r.sampled;
    return true;
}

void TkLlmBatcher::loop() {
    std::vector<int32_t> sq, ps, tk, am;
    std::vector<const uint32_t*> masks;
    std::vector<TkSampleRow> samps;
    struct Taken { Request* r; int take, last_row; };   /* `take` rows of a request end at row `last_row` of the pass */
    struct AheadRow { int slot, row; };                  /* a sequence's run-ahead row and where it sits in the pass */
    std::vector<Taken> in_pass;
    std::vector<Request*> completing;
    std::vector<AheadRow> ahead_rows;
    auto any_planned = [&] {
        for (const Ahead& a : ahead_) if (a.st == Ahead::PLANNED) return true;
        return false;
    };
    const auto t_origin = std::chrono::steady_clock::now();
    auto now_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_origin).count(); };
    for (;;) {
        double t_woke = 0.0;
        in_pass.clear(); completing.clear(); ahead_rows.clear();
        sq.clear(); ps.clear(); tk.clear(); masks.clear(); samps.clear();
        bool any_mask = false, any_samp = false;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return stop_ || !queue_.empty() || any_planned(); });
            t_woke = trace_path_.empty() ? 0.0 : now_ms();
            if (stop_) {
                for (Request* r : queue_) { r->ok = false; r->error = "scheduler stopped"; r->finished = true; r->cv.notify_all(); }
                queue_.clear();
                for (Ahead& a : ahead_)
                    if (a.waiter) { a.waiter->ok = false; a.waiter->error = "scheduler stopped"; a.waiter->finished = true; a.waiter->cv.notify_all(); a.waiter = nullptr; }
                return;
            }
            /* the owners of the requests the last pass finished WITHOUT a run-ahead row (masked steps) are about to ask for their next
             * token: give them a moment so that K runners decoding in lock step share every pass (a late one simply rides the next pass) */
            if (queue_.size() < expect_ && !any_planned()) {
                /* the window grows with the number of owners expected back (waking K host threads takes time) and stays well under the cost
                 * of the pass it fills: 200 us + 8 us per expected request, at most 2.5 ms (a 256-row pass is ~8 ms, a 16-row pass ~2.3 ms) */
                const int64_t us = std::min<int64_t>(2500, 200 + 8 * (int64_t)expect_);
                cv_.wait_for(lk, std::chrono::microseconds(us), [&] { return stop_ || queue_.size() >= expect_; });
            }
            /* FIFO: a request contributes as many of its remaining rows as the pass still holds, then the run-ahead rows.  A grammar-masked
             * request samples under its own token mask, which the arg max kernel takes per ROW: masked and unmasked requests share passes.
             * A row's result does not depend on its place in a pass or on the pass it rides.
             * TK_MI355X_BATCHER_DECODE_FIRST=1 (measured, not the default: profiles/r06_batcher_trace.txt) takes the rows that end in a sampled
             * token first — one-row requests and run-ahead rows — so that a decoding sequence rides the passes that carry other sequences'
             * prompt rows instead of waiting behind them: K cortices driven in lock step gain little (their prompts and their decode steps
             * arrive together whatever the order) and a pass that mixes one-row sequences with prompt chunks pays for 16-row attention
             * tiles that hold one row each (256-row pass: 8.7 -> 9.6 .. 12.3 ms). */
            auto add_rows = [&](Request* r, int take) {
                for (int i = 0; i < take; ++i) {
                    sq.push_back(r->slot);
                    ps.push_back(r->pos0 + r->done_rows + i);
                    tk.push_back(r->toks[r->done_rows + i]);
                    masks.push_back(nullptr);
                    samps.push_back(TkSampleRow{});
                }
                in_pass.push_back(Taken{r, take, (int)sq.size() - 1});
                if (r->done_rows + take == r->n) {
                    completing.push_back(r);
                    masks.back() = r->mask; /* the row that is sampled */
                    samps.back() = r->samp;
                    any_mask = any_mask || r->mask != nullptr;
                    any_samp = any_samp || r->samp.temp > 0.0f;
                }
            };
            const bool decode_first = decode_first_;
            if (decode_first)
                for (Request* r : queue_) {
                    if ((int)sq.size() >= TK_MAX_ROWS) break;
                    if (r->n - r->done_rows == 1) add_rows(r, 1);
                }
            auto add_ahead = [&] {
                for (size_t s = 0; s < ahead_.size() && (int)sq.size() < TK_MAX_ROWS; ++s) {
                    Ahead& a = ahead_[s];
                    if (a.st != Ahead::PLANNED) continue;
                    a.st = Ahead::INFLIGHT;
                    sq.push_back((int32_t)s);
                    ps.push_back(a.pos);
                    tk.push_back(a.tok);
                    masks.push_back(nullptr);
                    samps.push_back(TkSampleRow{});
                    ahead_rows.push_back(AheadRow{(int)s, (int)sq.size() - 1});
                }
            };
            if (decode_first) add_ahead(); /* one per sequence whose owner holds the id they feed */
            for (Request* r : queue_) {
                if ((int)sq.size() >= TK_MAX_ROWS) break;
                if (decode_first && r->n - r->done_rows == 1) continue; /* taken above */
                add_rows(r, std::min(r->n - r->done_rows, TK_MAX_ROWS - (int)sq.size()));
            }
            if (!decode_first) add_ahead();
        }
        const int nrows = (int)sq.size();
        if (nrows == 0) continue;
        am.assign((size_t)nrows, -1);
        const double t_formed = trace_path_.empty() ? 0.0 : now_ms();
        const bool head = !completing.empty() || !ahead_rows.empty();
        const bool ok = session_.forward(nrows, sq.data(), ps.data(), tk.data(), nullptr, head ? am.data() : nullptr, head, head && any_mask ? masks.data() : nullptr,
                                        head && any_samp ? samps.data() : nullptr);
        {
            std::lock_guard<std::mutex> lk(mu_);
            passes_++;
            if (nrows > max_rows_) max_rows_ = nrows;
            if (!ok) last_error_ = session_.error;
            size_t held = 0; /* completing requests that got no run-ahead row: their owners are waited for */
            for (const Taken& t : in_pass) {
                Request* r = t.r;
                const int take = t.take, row = t.last_row + 1;
                rows_ += (uint64_t)take;
                r->done_rows += take;
                if (!ok) { r->ok = false; r->error = session_.error; r->done_rows = r->n; }
                if (r->done_rows == r->n) {
                    r->sampled = ok ? am[(size_t)row - 1] : -1;
                    for (auto it = queue_.begin(); it != queue_.end(); ++it)
                        if (*it == r) { queue_.erase(it); break; }
                    if (ok) plan_ahead(r->slot, r->pos0 + r->n - 1, r->sampled, r->mask != nullptr || r->samp.temp > 0.0f);
                    if (!ok || ahead_[(size_t)r->slot].st != Ahead::PLANNED) held++;
                    r->finished = true;
                    r->cv.notify_all();
                }
            }
            for (const AheadRow& ar : ahead_rows) {
                const int s = ar.slot;
                Ahead& a = ahead_[(size_t)s];
                const int32_t got = ok ? am[(size_t)ar.row] : -1;
                if (a.discard) { wasted_++; a = Ahead(); continue; }
                if (a.waiter) { /* the owner is already waiting for exactly this row */
                    Request* w = a.waiter;
                    const int pos = a.pos;
                    a = Ahead();
                    w->ok = ok;
                    if (!ok) w->error = session_.error;
                    w->sampled = got;
                    rows_++;
                    if (ok) plan_ahead(s, pos, got, false);
                    w->finished = true;
                    w->cv.notify_all();
                } else {
                    a.st = Ahead::DONE;
                    a.sampled = got;
                    a.ok = ok;
                }
            }
            expect_ = held;
            if (!trace_path_.empty())
                trace_.push_back(PassTrace{t_woke, t_formed, now_ms(), nrows, (int)ahead_rows.size(), (int)completing.size(), (int)queue_.size(), session_.n_captures, session_.capture_ms});
        }
    }
}
