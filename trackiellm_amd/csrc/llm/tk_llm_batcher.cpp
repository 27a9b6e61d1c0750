#include "tk_llm_batcher.h"

#include <algorithm>
#include <chrono>

TkLlmBatcher::~TkLlmBatcher() {
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
    }
    cv_.notify_all();
    if (worker_.joinable()) worker_.join();
}

bool TkLlmBatcher::init(TkLlmModel* model, int slots, int n_ctx, std::string* err) {
    if (slots < 1) slots = 1;
    if (slots > TK_MAX_ROWS) slots = TK_MAX_ROWS; /* one decode row per runner must fit one pass */
    if (!session_.init(model, slots, n_ctx)) { *err = session_.error; return false; }
    n_ctx_ = n_ctx;
    slot_used_.assign((size_t)slots, 0);
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
    if (slot >= 0 && slot < (int)slot_used_.size()) slot_used_[(size_t)slot] = 0;
}

void TkLlmBatcher::stats(uint64_t* passes, uint64_t* rows, int* max_rows) {
    std::lock_guard<std::mutex> lk(mu_);
    if (passes) *passes = passes_;
    if (rows) *rows = rows_;
    if (max_rows) *max_rows = max_rows_;
}

bool TkLlmBatcher::submit(int slot, int pos0, const int32_t* toks, int n, const uint32_t* mask, int32_t* sampled, std::string* err) {
    if (n <= 0) { *err = "nothing to feed"; return false; }
    if (slot < 0 || slot >= (int)slot_used_.size() || pos0 < 0 || pos0 + n > n_ctx_) { *err = "rows do not fit the context window"; return false; }
    Request r;
    r.slot = slot; r.pos0 = pos0; r.n = n; r.toks = toks; r.mask = mask;
    std::unique_lock<std::mutex> lk(mu_);
    if (stop_) { *err = "scheduler stopped"; return false; }
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
    std::vector<Request*> in_pass, completing;
    for (;;) {
        in_pass.clear(); completing.clear();
        sq.clear(); ps.clear(); tk.clear(); masks.clear();
        bool any_mask = false;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
            if (stop_) {
                for (Request* r : queue_) { r->ok = false; r->error = "scheduler stopped"; r->finished = true; r->cv.notify_all(); }
                queue_.clear();
                return;
            }
            /* the owners of the requests the last pass finished are about to ask for their next token: give them a moment so that K
             * runners decoding in lock step share every pass (a late one simply rides the next pass) */
            if (queue_.size() < expect_) {
                /* the window grows with the number of owners expected back (waking K host threads takes time) and stays well under the cost
                 * of the pass it fills: 200 us + 8 us per expected request, at most 2.5 ms (a 256-row pass is ~8 ms, a 16-row pass ~2.3 ms) */
                const int64_t us = std::min<int64_t>(2500, 200 + 8 * (int64_t)expect_);
                cv_.wait_for(lk, std::chrono::microseconds(us), [&] { return stop_ || queue_.size() >= expect_; });
            }
            /* FIFO; a request contributes as many of its remaining rows as the pass still holds.  A grammar-masked request samples
             * under its own token mask, which the arg max kernel takes per ROW: masked and unmasked requests share passes. */
            for (Request* r : queue_) {
                if ((int)sq.size() >= TK_MAX_ROWS) break;
                const int take = std::min(r->n - r->done_rows, TK_MAX_ROWS - (int)sq.size());
                for (int i = 0; i < take; ++i) {
                    sq.push_back(r->slot);
                    ps.push_back(r->pos0 + r->done_rows + i);
                    tk.push_back(r->toks[r->done_rows + i]);
                    masks.push_back(nullptr);
                }
                in_pass.push_back(r);
                if (r->done_rows + take == r->n) {
                    completing.push_back(r);
                    masks.back() = r->mask; /* the row that is sampled */
                    any_mask = any_mask || r->mask != nullptr;
                }
            }
        }
        const int nrows = (int)sq.size();
        am.assign((size_t)nrows, -1);
        const bool head = !completing.empty();
        const bool ok = session_.forward(nrows, sq.data(), ps.data(), tk.data(), nullptr, head ? am.data() : nullptr, head, head && any_mask ? masks.data() : nullptr);
        {
            std::lock_guard<std::mutex> lk(mu_);
            passes_++;
            rows_ += (uint64_t)nrows;
            if (nrows > max_rows_) max_rows_ = nrows;
            int row = 0;
            for (Request* r : in_pass) {
                const int take = std::min(r->n - r->done_rows, TK_MAX_ROWS - row);
                row += take;
                r->done_rows += take;
                if (!ok) { r->ok = false; r->error = session_.error; r->done_rows = r->n; }
                if (r->done_rows == r->n) {
                    r->sampled = ok ? am[(size_t)row - 1] : -1;
                    for (auto it = queue_.begin(); it != queue_.end(); ++it)
                        if (*it == r) { queue_.erase(it); break; }
                    r->finished = true;
                    r->cv.notify_all();
                }
            }
            expect_ = completing.size();
        }
    }
}
