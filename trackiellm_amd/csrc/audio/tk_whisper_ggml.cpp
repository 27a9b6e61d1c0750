/* tk_whisper_ggml.cpp — see tk_whisper_ggml.h */
#include "tk_whisper_ggml.h"

#include <stdexcept>

#include <stdio.h>
#include <string.h>

#include "../common/tk_exact_math.h"

namespace {
struct File {
    FILE* f = nullptr;
    ~File() { if (f) fclose(f); }
    bool rd(void* p, size_t n) { return fread(p, 1, n, f) == n; }
};
}  // namespace

bool TkWhisperGgml::is_ggml(const char* path) {
    File fl;
    fl.f = fopen(path, "rb");
    uint32_t magic = 0;
    return fl.f && fl.rd(&magic, 4) && magic == 0x67676d6cu;
}

bool TkWhisperGgml::open(const char* path) {
    try { /* file contents are untrusted: nothing may throw through the extern "C" callers */
        return open_checked(path);
    } catch (const std::exception& e) {
        error = std::string("corrupt whisper ggml file (") + e.what() + ")";
        return false;
    }
}

bool TkWhisperGgml::open_checked(const char* path) {
    path_ = path;
    tensors.clear();
    vocab.clear();
    File fl;
    fl.f = fopen(path, "rb");
    if (!fl.f) { error = std::string("cannot open ") + path; return false; }
    uint32_t magic = 0;
    int32_t h[11];
    if (!fl.rd(&magic, 4) || magic != 0x67676d6cu) { error = "not a whisper ggml file (magic)"; return false; }
    if (!fl.rd(h, sizeof h)) { error = "truncated header"; return false; }
    hp = TkWhisperHP{h[9], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[0]};
    ftype = h[10];
    if (hp.n_mels <= 0 || hp.n_mels > 256 || hp.n_audio_ctx <= 0 || hp.n_audio_ctx > 4096 || hp.n_audio_state <= 0 || hp.n_audio_state > 4096 ||
        hp.n_audio_head <= 0 || hp.n_audio_layer <= 0 || hp.n_audio_layer > 64 || hp.n_text_ctx <= 0 || hp.n_text_ctx > 4096 ||
        hp.n_text_state <= 0 || hp.n_text_state > 4096 || hp.n_text_head <= 0 || hp.n_text_layer <= 0 || hp.n_text_layer > 64 ||
        hp.n_vocab <= 0 || hp.n_vocab > (1 << 20)) { error = "implausible hyper-parameters"; return false; }
    int32_t n_mel = 0, n_fft = 0;
    if (!fl.rd(&n_mel, 4) || !fl.rd(&n_fft, 4) || n_mel != hp.n_mels || n_fft != TK_WH_NBINS) { error = "mel filter bank has the wrong shape"; return false; }
    mel_filters.resize((size_t)n_mel * n_fft);
    if (!fl.rd(mel_filters.data(), mel_filters.size() * 4)) { error = "truncated filter bank"; return false; }
    int32_t n_tok = 0;
    if (!fl.rd(&n_tok, 4) || n_tok < 0 || n_tok > hp.n_vocab) { error = "bad vocabulary size"; return false; }
    vocab.resize((size_t)n_tok);
    for (int i = 0; i < n_tok; ++i) {
        uint32_t len = 0;
        if (!fl.rd(&len, 4) || len > 4096) { error = "bad vocabulary entry"; return false; }
        vocab[i].resize(len);
        if (len && !fl.rd(&vocab[i][0], len)) { error = "truncated vocabulary"; return false; }
    }
    fseek(fl.f, 0, SEEK_END);
    const long fsize = ftell(fl.f);
    long pos = 0;
    {   /* back to the end of the vocabulary */
        long p = 4 + (long)sizeof h + 8 + (long)mel_filters.size() * 4 + 4;
        for (const auto& s : vocab) p += 4 + (long)s.size();
        pos = p;
    }
    while (pos < fsize) {
        fseek(fl.f, pos, SEEK_SET);
        int32_t d[3];
        if (!fl.rd(d, sizeof d)) { error = "truncated tensor header"; return false; }
        TkWhisperGgmlTensor t;
        t.n_dims = d[0];
        t.type = d[2];
        if (t.n_dims < 1 || t.n_dims > 4 || d[1] <= 0 || d[1] > 256) { error = "bad tensor header"; return false; }
        if (t.type != 0 && t.type != 1) { error = "quantised whisper checkpoints (ggml type " + std::to_string(t.type) + ") are not supported: use the f16 / f32 file"; return false; }
        t.count = 1;
        for (int i = 0; i < t.n_dims; ++i) {
            int32_t ne = 0;
            if (!fl.rd(&ne, 4) || ne <= 0) { error = "bad tensor shape"; return false; }
            t.ne[i] = ne;
            if (t.count > (int64_t)fsize / ne) { error = "tensor element count exceeds the file size"; return false; } /* keeps count * ne from wrapping */
            t.count *= ne;
        }
        t.name.resize((size_t)d[1]);
        if (!fl.rd(&t.name[0], (size_t)d[1])) { error = "truncated tensor name"; return false; }
        t.offset = ftell(fl.f);
        const int64_t bytes = t.count * (t.type == 1 ? 2 : 4); /* count <= fsize: cannot wrap */
        if (t.offset < 0 || t.offset > fsize || bytes > (int64_t)fsize - t.offset) { error = "tensor " + t.name + " runs past the end of the file"; return false; }
        pos = (long)(t.offset + bytes);
        tensors.push_back(t);
    }
    return true;
}

bool TkWhisperGgml::read(const TkWhManifest& man, int idx, std::vector<float>* out, bool* found) {
    const TkWhTensor& want = man.t[idx];
    *found = false;
    if (idx == man.melw) {
        *out = mel_filters;
        *found = true;
        return true;
    }
    const TkWhisperGgmlTensor* t = nullptr;
    for (const auto& c : tensors)
        if (c.name == want.name) { t = &c; break; }
    if (!t) return false;
    *found = true;
    if (t->count != want.rows * want.cols) { error = "tensor " + want.name + " has " + std::to_string(t->count) + " elements, expected " + std::to_string(want.rows * want.cols); return false; }
    File fl;
    fl.f = fopen(path_.c_str(), "rb");
    if (!fl.f) { error = "cannot reopen " + path_; return false; }
    fseek(fl.f, (long)t->offset, SEEK_SET);
    std::vector<float> raw((size_t)t->count);
    if (t->type == 0) {
        if (!fl.rd(raw.data(), raw.size() * 4)) { error = "read error"; return false; }
    } else {
        std::vector<uint16_t> hbuf((size_t)t->count);
        if (!fl.rd(hbuf.data(), hbuf.size() * 2)) { error = "read error"; return false; }
        for (size_t i = 0; i < hbuf.size(); ++i) raw[i] = tk_f16_to_f32(hbuf[i]);
    }
    const bool conv = t->n_dims == 3; /* [out][in][3] stored with ne = {3, in, out} */
    if (conv) {
        const int64_t taps = t->ne[0], cin = t->ne[1], cout = t->ne[2];
        if (taps != 3 || cout != want.rows || cin * 3 != want.cols) { error = "conv tensor " + want.name + " has the wrong shape"; return false; }
        out->resize(raw.size());
        for (int64_t o = 0; o < cout; ++o)
            for (int64_t c = 0; c < cin; ++c)
                for (int64_t k = 0; k < 3; ++k) (*out)[(size_t)(o * 3 * cin + k * cin + c)] = raw[(size_t)((o * cin + c) * 3 + k)];
    } else {
        if (t->n_dims == 2 && !(t->ne[0] == want.cols && t->ne[1] == want.rows) && !(t->ne[0] == 1 && t->ne[1] == want.cols)) {
            error = "tensor " + want.name + " has the wrong shape";
            return false;
        }
        out->swap(raw);
    }
    return true;
}
