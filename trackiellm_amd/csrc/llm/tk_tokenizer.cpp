#include "tk_tokenizer.h"

#include <stdio.h>

#include <queue>

void TkTokenizer::init_bytes(int vocab_size) {
    vocab = vocab_size;
    tokens_.clear();
    index_.clear();
}

void TkTokenizer::init_spm(const std::vector<std::string>& tokens, const std::vector<float>& scores, const std::vector<int32_t>& types,
                           int bos_id, int eos_id) {
    tokens_ = tokens;
    scores_ = scores;
    types_ = types;
    scores_.resize(tokens_.size(), 0.0f);
    types_.resize(tokens_.size(), 1);
    vocab = (int)tokens_.size();
    bos = bos_id;
    eos = eos_id;
    index_.clear();
    for (int i = 0; i < vocab; ++i) index_.emplace(tokens_[i], i);
}

int TkTokenizer::byte_token(uint8_t b) const {
    if (tokens_.empty()) return 3 + b;
    char buf[8];
    snprintf(buf, sizeof buf, "<0x%02X>", b);
    auto it = index_.find(buf);
    return it == index_.end() ? unk : it->second;
}

std::vector<int32_t> TkTokenizer::encode(const std::string& text, bool add_bos) const {
    std::vector<int32_t> out;
    if (add_bos) out.push_back(bos);
    if (tokens_.empty()) {
        for (unsigned char c : text) {
            int id = 3 + c;
            out.push_back(id < vocab ? id : unk);
        }
        return out;
    }
    /* llama.cpp (and the sentencepiece library) turn EMPTY text into no tokens at all — [bos] alone with add_bos — not into the dummy-prefix
     * piece: the space prefix belongs to the first text fragment, and empty text has none (include/tk/ABI_NOTES.md, "Tokenizer") */
    if (text.empty()) return out;
    /* SentencePiece-BPE, llama flavour: prefix space, spaces -> U+2581 */
    std::string s = " " + text;
    std::string norm;
    for (char c : s) {
        if (c == ' ') norm += "\xE2\x96\x81";
        else norm += c;
    }
    struct Sym { int prev, next; size_t off, len; };
    std::vector<Sym> syms;
    for (size_t i = 0; i < norm.size();) {
        unsigned char c = (unsigned char)norm[i];
        static const uint8_t len_by_high_nibble[16] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 3, 4}; /* llama.cpp's utf8_len: by the lead byte's high nibble, also for malformed input */
        size_t n = len_by_high_nibble[c >> 4];
        if (i + n > norm.size()) n = norm.size() - i;
        syms.push_back(Sym{(int)syms.size() - 1, (int)syms.size() + 1, i, n});
        i += n;
    }
    if (!syms.empty()) syms.back().next = -1;
    struct Bigram { int left, right; float score; size_t size; };
    auto cmp = [](const Bigram& a, const Bigram& b) { return a.score < b.score || (a.score == b.score && a.left > b.left); };
    std::priority_queue<Bigram, std::vector<Bigram>, decltype(cmp)> pq(cmp);
    auto try_add = [&](int l, int r) {
        if (l < 0 || r < 0) return;
        std::string t = norm.substr(syms[l].off, syms[l].len + syms[r].len);
        auto it = index_.find(t);
        if (it == index_.end()) return;
        pq.push(Bigram{l, r, scores_[it->second], t.size()});
    };
    for (int i = 1; i < (int)syms.size(); ++i) try_add(i - 1, i);
    while (!pq.empty()) {
        Bigram b = pq.top();
        pq.pop();
        Sym& L = syms[b.left];
        Sym& R = syms[b.right];
        if (L.len == 0 || R.len == 0 || L.len + R.len != b.size) continue; /* stale entry */
        L.len += R.len;
        R.len = 0;
        L.next = R.next;
        if (R.next >= 0) syms[R.next].prev = b.left;
        try_add(L.prev, b.left);
        try_add(b.left, L.next);
    }
    for (int i = 0; i != -1 && i < (int)syms.size(); i = syms[i].next) {
        if (syms[i].len == 0) continue;
        std::string t = norm.substr(syms[i].off, syms[i].len);
        auto it = index_.find(t);
        if (it != index_.end()) out.push_back(it->second);
        else
            for (unsigned char c : t) out.push_back(byte_token(c));
    }
    return out;
}

std::string TkTokenizer::piece(int32_t id) const {
    if (tokens_.empty()) {
        if (id >= 3 && id < 259) return std::string(1, (char)(id - 3));
        if (id == bos || id == eos || id == unk) return std::string();
        char buf[24];
        snprintf(buf, sizeof buf, " t%d", id);
        return buf;
    }
    if (id < 0 || id >= vocab) return std::string();
    if (types_[id] == 6) { /* byte token <0xXX> */
        unsigned v = 0;
        if (sscanf(tokens_[id].c_str(), "<0x%02X>", &v) == 1) return std::string(1, (char)v);
    }
    if (types_[id] == 3 || types_[id] == 2) return std::string(); /* control / unknown */
    std::string out;
    const std::string& t = tokens_[id];
    for (size_t i = 0; i < t.size();) {
        if (i + 2 < t.size() && (unsigned char)t[i] == 0xE2 && (unsigned char)t[i + 1] == 0x96 && (unsigned char)t[i + 2] == 0x81) {
            out += ' ';
            i += 3;
        } else {
            out += t[i++];
        }
    }
    return out;
}
