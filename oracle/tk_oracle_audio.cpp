/*
 * tk_oracle_audio.cpp — TEST INFRASTRUCTURE ONLY (oracle) for the ASR / VAD streams.
 *
 * PARITY UNPINNED against the reference for the network arithmetic: whisper.cpp is #included but not shipped,
 * declared or pinned (src/audio/tk_asr_whisper.c:24) and the Silero ONNX graph is absent.  The
 * Whisper graph (trackiellm_amd/csrc/common/tk_whisper_graph.h) is walked here with scalar CPU ops and pinned against
 * HF transformers' WhisperModel + WhisperFeatureExtractor (tests/golden/make_audio_golden.py).
 * Restated from reference sources: the VAD state machine (src/sensors/tk_vad_silero.c:283-322) and its windowing
 * (:327-390, 30 ms window, 10 ms hop, time advances by the window length), s16 -> f32 by /32768 (:78-82).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../trackiellm_amd/csrc/common/tk_whisper_graph.h"

extern "C" int32_t orc_sample_row(const float* logits, int vocab, const uint32_t* allow, float temp, int32_t top_k, float top_p, float min_p, uint64_t seed,
                                  uint32_t counter); /* tk_oracle_llm.cpp */
extern "C" void orc_gemm(const float* A, const float* B, float* C, const float* bias, const float* residual, int M, int N, int K, int lda, int ldb,
                         int ldc, int ldr, int b_kn, int act, float alpha);

static float sum256(const float* partial) {
    float w[4];
    for (int wv = 0; wv < 4; ++wv) {
        float p[64];
        for (int i = 0; i < 64; ++i) p[i] = partial[wv * 64 + i];
        for (int s = 32; s >= 1; s >>= 1)
            for (int i = 0; i < s; ++i) p[i] = p[i] + p[i + s];
        w[wv] = p[0];
    }
    return ((w[0] + w[1]) + w[2]) + w[3];
}

extern "C" int32_t orc_sample_row(const float* logits, int vocab, const uint32_t* allow, float temp, int32_t top_k, float top_p, float min_p, uint64_t seed, uint32_t counter);

/* 1024 strided chains of `term(i)` over i in [0, n) with keep(i), each wave of 64 chains joined by the xor butterfly (32 .. 1), then waves 0 .. 15 in
 * order: the order k_pick_rows / k_pick_rows_filtered sum in */
template <typename Keep, typename Term>
static float chain_sum_1024(int n, Keep keep, Term term) {
    float chain[1024];
    for (int t = 0; t < 1024; ++t) {
        float s = 0.0f;
        for (int i = t; i < n; i += 1024)
            if (keep(i)) s = s + term(i);
        chain[t] = s;
    }
    float S = 0.0f;
    for (int w = 0; w < 16; ++w) {
        float a[64], nx[64];
        for (int j = 0; j < 64; ++j) a[j] = chain[w * 64 + j];
        for (int sft = 32; sft >= 1; sft >>= 1) {
            for (int j = 0; j < 64; ++j) nx[j] = a[j] + a[j ^ sft];
            for (int j = 0; j < 64; ++j) a[j] = nx[j];
        }
        S = w == 0 ? a[0] : S + a[0];
    }
    return S;
}

/* One sampled position of whisper.cpp's decoder under the reference's parameters (TEST INFRASTRUCTURE, like everything under oracle/).  Follows
 * whisper.cpp's published whisper_process_logits and the token bookkeeping of whisper_full_with_state's decode loop — whisper.cpp is a dependency
 * the reference calls (src/audio/tk_asr_whisper.c:89-110,142-147) but does not vendor or pin: PARITY UNPINNED against it.  Parameters as the
 * reference's wrapper leaves them: suppress_blank = false, suppress_non_speech_tokens = true (in `suppress`, with the special tokens
 * whisper.cpp always masks: <|notimestamps|>, <|startoftranscript|>, <|nospeech|>, <|startoflm|>, <|startofprev|>, task and language tokens),
 * no_timestamps = false, max_initial_ts = 1.0 (tid0 = 50), single_segment = false, max_tokens = 0, seek = 0.
 * state[8] = {tokens sampled, last was a timestamp, the one before was, has_ts, seek_delta, result_len, status (0 running, 1 completed, 2 failed),
 * seek_end}.  Returns the token (eot for a row that stands still), *logprob = its log-probability under the filtered distribution.
 * Stated deviations: the sums run in this build's canonical order (1024 chains) instead of index order, exp / log are the build's exact-math
 * sequences, the greedy pick is the first index of the largest LOGIT (whisper.cpp takes the first index of the largest exp(logprob): equal unless
 * two allowed logits differ by less than an ulp of their probability), a draw at temperature > 0 is the build's canonical sampler over the 64
 * largest allowed logits (whisper.cpp: std::discrete_distribution over all). */
extern "C" int32_t orc_whisper_filter_pick(const float* x, int cols, const uint8_t* suppress, int32_t* st, int32_t beg, int32_t eot, int32_t tid0, float temperature,
                                           uint64_t seed, uint32_t counter, float* logprob) {
    if (st[6] != 0) { if (logprob) *logprob = 0.0f; return eot; }
    const int n_tok = st[0];
    const bool last_ts = n_tok > 0 && st[1] != 0, prev_ts = n_tok < 2 || st[2] != 0, has_ts = st[3] != 0;
    const int seek_delta = st[4];
    std::vector<uint8_t> ok((size_t)cols);
    for (int i = 0; i < cols; ++i) {
        bool a = suppress[i] == 0;
        if (last_ts) { /* timestamps have to appear in pairs, except directly before EOT */
            if (prev_ts) { if (i >= beg) a = false; }
            else if (i < eot) a = false;
        }
        if (n_tok == 0 && tid0 >= 0 && i > beg + tid0) a = false;      /* the initial timestamp cannot be larger than max_initial_ts */
        if (has_ts && i >= beg && i < beg + seek_delta / 2) a = false; /* condition timestamp tokens to be increasing */
        ok[(size_t)i] = a ? 1 : 0;
    }
    const bool hot = temperature > 0.0f;
    std::vector<float> z((size_t)cols);
    for (int i = 0; i < cols; ++i) z[(size_t)i] = hot ? tk_divf(x[i], temperature) : x[i];
    float m_all = -INFINITY, m_tx = -INFINITY, m_ts = -INFINITY;
    for (int i = 0; i < cols; ++i)
        if (ok[(size_t)i]) {
            m_all = tk_fmaxf(m_all, z[(size_t)i]);
            if (i < beg) m_tx = tk_fmaxf(m_tx, z[(size_t)i]); else m_ts = tk_fmaxf(m_ts, z[(size_t)i]);
        }
    const float S = chain_sum_1024(cols, [&](int i) { return ok[(size_t)i] != 0; }, [&](int i) { return tk_expf(z[(size_t)i] - m_all); });
    const float logS = tk_logf(S);
    auto lp = [&](int i) { return (z[(size_t)i] - m_all) - logS; };
    /* if sum of probability over timestamps is above any other token, sample timestamp */
    const float lp_mts = (m_ts - m_all) - logS, lp_mtx = (m_tx - m_all) - logS;
    const float S_ts = chain_sum_1024(cols, [&](int i) { return i >= beg && ok[(size_t)i] != 0; }, [&](int i) { return tk_expf(lp(i) - lp_mts); });
    const float lp_ts = S_ts > 0.0f ? tk_logf(S_ts) + lp_mts : -INFINITY;
    if (lp_ts > lp_mtx)
        for (int i = 0; i < beg && i < cols; ++i) ok[(size_t)i] = 0;
    int tok = -1;
    if (!hot) {
        for (int i = 0; i < cols; ++i)
            if (ok[(size_t)i] && (tok < 0 || x[i] > x[tok])) tok = i;
    } else {
        std::vector<uint32_t> allow((size_t)(cols + 31) / 32, 0u);
        for (int i = 0; i < cols; ++i)
            if (ok[(size_t)i]) allow[(size_t)i >> 5] |= 1u << (i & 31);
        tok = orc_sample_row(x, cols, allow.data(), temperature, 0, 1.0f, 0.0f, seed, counter);
    }
    if (logprob) *logprob = lp(tok);
    /* whisper_full_with_state, "update sliding window" / "end of segment", i = the token's index in the sequence */
    const int i = n_tok, seek_end = st[7];
    int hts = has_ts ? 1 : 0, sd = seek_delta, rl = st[5], status = 0;
    if (tok > beg) {
        const int sd_new = 2 * (tok - beg);
        if (hts && sd > sd_new && rl < i) status = 2; /* do not allow to go back in time */
        else { sd = sd_new; rl = i + 1; hts = 1; }
    }
    if (status == 0 && (tok == eot || (hts && sd + 100 >= seek_end))) {
        if (rl == 0) {
            if (sd + 100 >= seek_end) rl = i + 1;
            else status = 2;
        }
        if (status == 0) status = 1;
    }
    const int was_last = last_ts ? 1 : 0;
    st[0] = i + 1; st[1] = tok >= beg ? 1 : 0; st[2] = was_last; st[3] = hts; st[4] = sd; st[5] = rl; st[6] = status;
    return tok;
}

struct CpuAudioOps {
    std::vector<std::vector<float>> bufs;
    float* alloc(size_t n) { bufs.emplace_back(n ? n : 1, 0.0f); return bufs.back().data(); }
    int32_t* alloc_i32(size_t n) { return (int32_t*)alloc(n); }
    void gemm(const TkGemm& g) {
        const int nb = g.batch > 0 ? g.batch : 1;
        for (int z = 0; z < nb; ++z) {
            int64_t oA, oB, oC, oR;
            if (g.batch_inner > 0) {
                const int zo = z / g.batch_inner, zi = z % g.batch_inner;
                oA = zo * g.sA2 + zi * g.sA; oB = zo * g.sB2 + zi * g.sB; oC = zo * g.sC2 + zi * g.sC; oR = zo * g.sR2 + zi * g.sR;
            } else { oA = z * g.sA; oB = z * g.sB; oC = z * g.sC; oR = z * g.sR; }
            orc_gemm(g.A + oA, g.B + oB, g.C + oC, g.bias, g.residual ? g.residual + oR : nullptr, g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr,
                     g.b_kn, g.act, g.alpha);
        }
    }
    void im2col1d(const float* x, int B, int T, int C, int ldx, int kw, int stride, int pad, float* col) {
        const int To = (T + 2 * pad - kw) / stride + 1;
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < To; ++t)
                for (int kx = 0; kx < kw; ++kx) {
                    const int it = t * stride + kx - pad;
                    for (int c = 0; c < C; ++c)
                        col[(((int64_t)b * To + t) * kw + kx) * C + c] = (it >= 0 && it < T) ? x[((int64_t)b * T + it) * ldx + c] : 0.0f;
                }
    }
    void layernorm(const float* x, int rows, int D, const float* w, const float* b, float* y) {
        for (int r = 0; r < rows; ++r) {
            const float* xr = x + (int64_t)r * D;
            float part[256];
            for (int t = 0; t < 256; ++t) { float s = 0.0f; for (int i = t; i < D; i += 256) s = s + xr[i]; part[t] = s; }
            const float mean = tk_divf(sum256(part), (float)D);
            for (int t = 0; t < 256; ++t) { float q = 0.0f; for (int i = t; i < D; i += 256) { const float d = xr[i] - mean; q = tk_fmaf(d, d, q); } part[t] = q; }
            const float var = tk_divf(sum256(part), (float)D);
            const float rstd = tk_divf(1.0f, tk_sqrtf(var + TK_WH_LN_EPS));
            for (int i = 0; i < D; ++i) y[(int64_t)r * D + i] = ((xr[i] - mean) * rstd) * w[i] + b[i];
        }
    }
    bool attend1(const float*, const float*, const float*, float*, int, int, int64_t, int64_t, int, int) { return false; }
    bool attend_fused(const float*, const float*, const float*, float*, int, int, int, int64_t, int64_t, int, int) { return false; }
    void softmax_rows(float* x, int rows, int cols, int ld) {
        for (int r = 0; r < rows; ++r) {
            float* xr = x + (int64_t)r * ld;
            float m = -INFINITY;
            for (int i = 0; i < cols; ++i) m = tk_fmaxf(m, xr[i]);
            float part[256];
            for (int t = 0; t < 256; ++t) {
                float s = 0.0f;
                for (int i = t; i < cols; i += 256) { const float e = tk_expf(xr[i] - m); xr[i] = e; s = s + e; }
                part[t] = s;
            }
            const float tot = sum256(part);
            for (int i = 0; i < cols; ++i) xr[i] = tk_divf(xr[i], tot);
        }
    }
    void add_rows(float* x, const float* add, int rows, int D, int add_rows) {
        for (int64_t r = 0; r < rows; ++r)
            for (int i = 0; i < D; ++i) x[r * D + i] = x[r * D + i] + add[(r % add_rows) * D + i];
    }
    void embed_rows(const float* table, const float* pos, const int32_t* idx, const int32_t* pos_idx, int rows, int D, float* out) {
        for (int r = 0; r < rows; ++r)
            for (int i = 0; i < D; ++i) out[(int64_t)r * D + i] = table[(int64_t)idx[r] * D + i] + pos[(int64_t)pos_idx[r] * D + i];
    }
    /* whisper.cpp's decoding policy (the reference arms it: src/audio/tk_asr_whisper.c:126-138), restating k_pick_rows of
     * csrc/nn/tk_nn_kernels.hip: token by temperature (0 = first index of the maximum; > 0 = one draw of the canonical sampler over the 64
     * largest logits, counter = position x rows + row) and its log-probability under softmax(l / temperature) over the whole vocabulary */
    /* the reference-parameter decode (orc_whisper_filter_pick below): logits of the sampled positions go through whisper.cpp's filters */
    bool filt_on = false;
    int filt_first_step = 0;
    const uint8_t* filt_suppress = nullptr;
    int32_t* filt_state = nullptr; /* [rows][8] */
    int32_t filt_beg = 0, filt_eot = 0, filt_tid0 = 50;
    bool pick_on = false;
    float pick_temp = 0.0f;
    uint64_t pick_seed = 0;
    float* pick_lp = nullptr; /* [positions][rows] */
    int pick_step = 0;
    void argmax_rows(const float* x, int rows, int cols, int ld, int32_t* out) {
        if (filt_on) {
            if (pick_step >= filt_first_step)
                for (int r = 0; r < rows; ++r) {
                    float lp = 0.0f;
                    out[r] = orc_whisper_filter_pick(x + (int64_t)r * ld, cols, filt_suppress, filt_state + (int64_t)r * 8, filt_beg, filt_eot, filt_tid0, pick_temp, pick_seed,
                                                     (uint32_t)(pick_step * rows + r), &lp);
                    if (pick_lp) pick_lp[(int64_t)pick_step * rows + r] = lp;
                }
            pick_step++;
            return;
        }
        for (int r = 0; r < rows; ++r) {
            const float* xr = x + (int64_t)r * ld;
            int best = 0;
            for (int i = 1; i < cols; ++i) if (xr[i] > xr[best]) best = i;
            out[r] = best;
            if (!pick_on) continue;
            const float mx = xr[best];
            if (pick_temp > 0.0f)
                out[r] = orc_sample_row(xr, cols, nullptr, pick_temp, 0, 1.0f, 0.0f, pick_seed, (uint32_t)(pick_step * rows + r));
            if (!pick_lp) continue;
            /* sum of exponentials: 1024 strided chains, each wave of 64 chains joined by the xor butterfly (32 .. 1), then waves 0 .. 15 in order */
            float chain[1024];
            for (int t = 0; t < 1024; ++t) {
                float s = 0.0f;
                for (int i = t; i < cols; i += 1024) s = s + tk_expf(pick_temp > 0.0f ? tk_divf(xr[i] - mx, pick_temp) : xr[i] - mx);
                chain[t] = s;
            }
            float S = 0.0f;
            for (int w = 0; w < 16; ++w) {
                float a[64], nx[64];
                for (int j = 0; j < 64; ++j) a[j] = chain[w * 64 + j];
                for (int sft = 32; sft >= 1; sft >>= 1) {
                    for (int j = 0; j < 64; ++j) nx[j] = a[j] + a[j ^ sft];
                    for (int j = 0; j < 64; ++j) a[j] = nx[j];
                }
                S = w == 0 ? a[0] : S + a[0];
            }
            const float z = pick_temp > 0.0f ? tk_divf(xr[out[r]] - mx, pick_temp) : xr[out[r]] - mx;
            pick_lp[(int64_t)pick_step * rows + r] = z - tk_logf(S);
        }
        if (pick_on) pick_step++;
    }
    void frames(const int16_t* pcm, int B, int n_samples, int pcm_stride, int n_total, int T, const float* window, float* out) {
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < T; ++t)
                for (int n = 0; n < TK_WH_NFFT; ++n) {
                    int j = t * TK_WH_HOP + n - TK_WH_NFFT / 2;
                    if (j < 0) j = -j;
                    if (j >= n_total) j = 2 * (n_total - 1) - j;
                    const float x = j < n_samples ? (float)pcm[(int64_t)b * pcm_stride + j] / 32768.0f : 0.0f;
                    out[((int64_t)b * T + t) * TK_WH_NFFT + n] = x * window[n];
                }
    }
    void power(const float* ri, int rows, int nb, float* out) {
        for (int64_t r = 0; r < rows; ++r)
            for (int f = 0; f < nb; ++f) { const float re = ri[r * 2 * nb + f], im = ri[r * 2 * nb + nb + f]; out[r * nb + f] = tk_fmaf(re, re, im * im); }
    }
    void logmel_finish(float* mel, int B, int per_b) {
        for (int b = 0; b < B; ++b) {
            float* m = mel + (int64_t)b * per_b;
            float mx = -INFINITY;
            for (int i = 0; i < per_b; ++i) { m[i] = tk_log10f(tk_fmaxf(m[i], 1e-10f)); mx = tk_fmaxf(mx, m[i]); }
            const float fl = mx - 8.0f;
            for (int i = 0; i < per_b; ++i) m[i] = (tk_fmaxf(m[i], fl) + 4.0f) * 0.25f;
        }
    }
};

struct orc_whisper {
    TkWhisperHP hp;
    TkWhManifest man;
    std::vector<std::vector<float>> w;
    std::vector<float*> wp;
};

extern "C" {

orc_whisper* orc_whisper_create(const TkWhisperHP* hp, uint64_t seed) {
    orc_whisper* m = new orc_whisper();
    m->hp = *hp;
    m->man = tk_whisper_manifest(*hp);
    m->w.resize(m->man.t.size());
    for (size_t i = 0; i < m->man.t.size(); ++i) {
        m->w[i].resize((size_t)m->man.t[i].rows * m->man.t[i].cols);
        tk_whisper_fill_tensor(*hp, m->man, (int)i, seed, m->w[i].data());
        m->wp.push_back(m->w[i].data());
    }
    return m;
}
void orc_whisper_destroy(orc_whisper* m) { delete m; }
int orc_whisper_tensor_count(orc_whisper* m) { return (int)m->man.t.size(); }
int64_t orc_whisper_tensor_info(orc_whisper* m, int i, char* name, int cap, int64_t* rows, int64_t* cols) {
    snprintf(name, cap, "%s", m->man.t[i].name.c_str());
    *rows = m->man.t[i].rows; *cols = m->man.t[i].cols;
    return *rows * *cols;
}
void orc_whisper_get_tensor(orc_whisper* m, int i, float* out) { memcpy(out, m->w[i].data(), m->w[i].size() * 4); }
/* replace tensor i (manifest order, rows x cols floats): lets a test run the oracle on the weights it wrote into a checkpoint file */
void orc_whisper_set_tensor(orc_whisper* m, int i, const float* in) { memcpy(m->w[i].data(), in, m->w[i].size() * 4); }

/* same contract as TkAsr::transcribe: prompt tokens then n_steps forced greedy tokens */
void orc_whisper_transcribe(orc_whisper* m, int B, const int16_t* pcm, int n_samples, const int32_t* prompt, int n_prompt, int n_steps,
                            int32_t* tokens_out, float* mel_out, float* enc_out, float* first_logits) {
    const TkWhisperHP& h = m->hp;
    CpuAudioOps ops;
    ops.bufs.reserve(4096);
    TkWhisperGraph<CpuAudioOps> g{h, m->man, m->wp.data()};
    float* ml = g.mel(ops, pcm, B, n_samples, n_samples);
    if (mel_out) memcpy(mel_out, ml, (size_t)B * h.n_frames() * h.n_mels * 4);
    float* enc = g.encode(ops, ml, B);
    if (enc_out) memcpy(enc_out, enc, (size_t)B * h.n_audio_ctx * h.n_audio_state * 4);
    if (n_steps <= 0) return;
    auto st = g.begin_decode(ops, enc, B);
    const int total = n_prompt + n_steps - 1;
    for (int p = 0; p < total; ++p) {
        for (int b = 0; b < B; ++b) { st.tok[b] = p < n_prompt ? prompt[p] : st.next[b]; st.pos[b] = p; }
        g.decode_step(ops, st, p);
        if (p >= n_prompt - 1) {
            const int step = p - (n_prompt - 1);
            if (step == 0 && first_logits) memcpy(first_logits, st.logits, (size_t)B * h.n_vocab * 4);
            for (int b = 0; b < B; ++b) tokens_out[(size_t)b * n_steps + step] = st.next[b];
        }
    }
}

/* same contract as TkAsr::transcribe_policy: the forced decode with the token picked by temperature; logprobs_out [B][n_steps] */
void orc_whisper_transcribe_policy(orc_whisper* m, int B, const int16_t* pcm, int n_samples, const int32_t* prompt, int n_prompt, int n_steps,
                                   float temperature, uint64_t seed, int32_t* tokens_out, float* logprobs_out) {
    const TkWhisperHP& h = m->hp;
    CpuAudioOps ops;
    ops.bufs.reserve(4096);
    TkWhisperGraph<CpuAudioOps> g{h, m->man, m->wp.data()};
    float* ml = g.mel(ops, pcm, B, n_samples, n_samples);
    float* enc = g.encode(ops, ml, B);
    auto st = g.begin_decode(ops, enc, B);
    const int total = n_prompt + n_steps - 1;
    std::vector<float> lp((size_t)total * B);
    ops.pick_on = true; ops.pick_temp = temperature; ops.pick_seed = seed; ops.pick_lp = lp.data(); ops.pick_step = 0;
    for (int p = 0; p < total; ++p) {
        for (int b = 0; b < B; ++b) { st.tok[b] = p < n_prompt ? prompt[p] : st.next[b]; st.pos[b] = p; }
        g.decode_step(ops, st, p);
        if (p >= n_prompt - 1) {
            const int step = p - (n_prompt - 1);
            for (int b = 0; b < B; ++b) {
                tokens_out[(size_t)b * n_steps + step] = st.next[b];
                if (logprobs_out) logprobs_out[(size_t)b * n_steps + step] = lp[(size_t)p * B + b];
            }
        }
    }
}

/* same contract as TkAsr::transcribe_ref: the decode under the reference's whisper.cpp parameters, at most n_steps tokens per utterance */
void orc_whisper_transcribe_ref(orc_whisper* m, int B, const int16_t* pcm, int n_samples, const int32_t* n_samples_row, const int32_t* prompt, int n_prompt, int n_steps,
                                float temperature, uint64_t seed, const uint8_t* suppress, int32_t token_beg, int32_t token_eot, int32_t* tokens_out, float* logprobs_out,
                                int32_t* result_len, int32_t* status) {
    const TkWhisperHP& h = m->hp;
    CpuAudioOps ops;
    ops.bufs.reserve(4096);
    TkWhisperGraph<CpuAudioOps> g{h, m->man, m->wp.data()};
    float* ml = g.mel(ops, pcm, B, n_samples, n_samples);
    float* enc = g.encode(ops, ml, B);
    auto st = g.begin_decode(ops, enc, B);
    const int total = n_prompt + n_steps - 1;
    std::vector<float> lp((size_t)total * B, 0.0f);
    std::vector<int32_t> state((size_t)B * 8, 0);
    for (int b = 0; b < B; ++b) state[(size_t)b * 8 + 7] = n_samples_row[b] / TK_WH_HOP;
    ops.filt_on = true; ops.filt_first_step = n_prompt - 1; ops.filt_suppress = suppress; ops.filt_state = state.data(); ops.filt_beg = token_beg; ops.filt_eot = token_eot;
    ops.filt_tid0 = 50;
    ops.pick_temp = temperature; ops.pick_seed = seed; ops.pick_lp = lp.data(); ops.pick_step = 0;
    for (int p = 0; p < total; ++p) {
        for (int b = 0; b < B; ++b) { st.tok[b] = p < n_prompt ? prompt[p] : st.next[b]; st.pos[b] = p; }
        g.decode_step(ops, st, p);
        if (p >= n_prompt - 1) {
            const int step = p - (n_prompt - 1);
            for (int b = 0; b < B; ++b) {
                tokens_out[(size_t)b * n_steps + step] = st.next[b];
                if (logprobs_out) logprobs_out[(size_t)b * n_steps + step] = lp[(size_t)p * B + b];
            }
        }
    }
    for (int b = 0; b < B; ++b) {
        const int32_t* sb = &state[(size_t)b * 8];
        if (status) status[b] = sb[6];
        if (result_len) result_len[b] = sb[6] == 0 ? sb[0] : sb[5];
    }
}

/* whisper.cpp's acceptance test of one decode (the temperature-fallback loop of whisper_full_with_state; whisper.cpp is a dependency the reference does
 * not vendor): length = up to and including the first end-of-text token (else n_steps); failed = mean log-probability below logprob_thold, or — with
 * more than 32 tokens — the entropy of the histogram of the last 32 tokens below entropy_thold.  Returns 1 when failed; *avg = the mean. */
int orc_whisper_decode_failed(const int32_t* toks, const float* lp, int n_steps, int32_t eot, float entropy_thold, float logprob_thold, float* avg) {
    int len = n_steps;
    for (int i = 0; i < n_steps; ++i)
        if (toks[i] == eot) { len = i + 1; break; }
    double sum = 0.0;
    for (int i = 0; i < len; ++i) sum += (double)lp[i];
    const float a = (float)(sum / (double)len);
    if (avg) *avg = a;
    int failed = a < logprob_thold;
    if (len > 32) {
        double ent = 0.0;
        for (int i = len - 32; i < len; ++i) {
            bool first = true;
            for (int j = len - 32; j < i; ++j) if (toks[j] == toks[i]) { first = false; break; }
            if (!first) continue;
            int cnt = 0;
            for (int j = i; j < len; ++j) cnt += toks[j] == toks[i];
            const double pr = cnt / 32.0;
            ent -= pr * log(pr);
        }
        if (ent < (double)entropy_thold) failed = 1;
    }
    return failed;
}

/* ---- VAD ---- */
void orc_vad_probabilities(uint64_t seed, int window, int hidden, const float* windows, int n, float* prob) {
    std::vector<float> w1((size_t)window * hidden), b1(hidden), w2(hidden), b2(1), hid((size_t)n * hidden);
    tk_vad_synth(seed, window, hidden, w1.data(), b1.data(), w2.data(), b2.data());
    orc_gemm(windows, w1.data(), hid.data(), b1.data(), nullptr, n, hidden, window, window, window, hidden, 0, 0, 0, 1.0f);
    for (int i = 0; i < n; ++i) {
        float acc = 0.0f;
        for (int k = 0; k < hidden; ++k) acc = tk_fmaf(tk_fmaxf(hid[(size_t)i * hidden + k], 0.0f), w2[k], acc);
        prob[i] = tk_sigmoidf(acc + b2[0]);
    }
}

typedef struct {
    float threshold, min_silence_ms, min_speech_ms;
    int active, triggered;
    float prob, silence_ms, speech_ms, since_event_ms;
} orc_vad_state_t;

/* one step of the reference state machine (tk_vad_silero.c:283-322); returns -1 none, 0 started, 1 ended */
int orc_vad_step(orc_vad_state_t* s, float probability, float dt_ms) {
    const int before = s->active;
    s->prob = probability;
    s->since_event_ms += dt_ms;
    if (probability >= s->threshold) {
        s->speech_ms += dt_ms;
        s->silence_ms = 0.0f;
        if (!s->active && s->speech_ms >= s->min_speech_ms && !s->triggered) { s->active = 1; s->triggered = 1; s->since_event_ms = 0.0f; }
    } else {
        s->silence_ms += dt_ms;
        s->speech_ms = 0.0f;
        if (s->active && s->silence_ms >= s->min_silence_ms) { s->active = 0; s->triggered = 0; s->since_event_ms = 0.0f; }
    }
    if (!before && s->active) return 0;
    if (before && !s->active) return 1;
    return -1;
}

} /* extern "C" */
