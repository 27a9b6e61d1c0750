/*
 * tk_whisper_graph.h — Whisper (tiny.en-class) ASR stream written once against an abstract Ops backend, so
 * the HIP engine and the CPU oracle execute the same sequence of primitive ops (GEMM descriptors, layer norms,
 * softmaxes).  The reference only calls whisper_full() (src/audio/tk_asr_whisper.c:142-147); whisper.cpp is
 * absent and un-pinned, so the network follows the published Whisper definition
 * (n_fft 400, hop 160, 80 slaney mel bins, 30 s window -> 1500 encoder positions, pre-LN blocks, tanh-GELU as
 * ggml evaluates it, tied output embedding) and is pinned against HF transformers' WhisperModel + feature
 * extractor in tests/golden/make_audio_golden.py.
 *
 * Ops concept (pointers are device pointers for the HIP Ops, host pointers for the oracle Ops):
 *   float*   alloc(size_t n);  int32_t* alloc_i32(size_t n);
 *   void gemm(const TkGemm&);
 *   void im2col1d(const float* x, int B, int T, int C, int ldx, int kw, int stride, int pad, float* col);
 *   void layernorm(const float* x, int rows, int D, const float* w, const float* b, float* y);
 *   void softmax_rows(float* x, int rows, int cols, int ld);
 *   bool attend1(...): optional fused form of attention() for one query row per sequence (same arithmetic); false = not provided
 *   bool attend_fused(...): optional one-kernel form of attention() for many query rows (the GPU's opt-in fast contraction: NOT the same
 *     arithmetic, ~1e-6 of scale off; the oracle and the exact path return false)
 *   void add_rows(float* x, const float* add, int rows, int D, int add_rows);
 *   void embed_rows(const float* table, const float* pos, const int32_t* idx, const int32_t* pos_idx, int rows, int D, float* out);
 *   void argmax_rows(const float* x, int rows, int cols, int ld, int32_t* out);
 *   void frames(const int16_t* pcm, int B, int n_samples, int pcm_stride, int n_total, int T, const float* window, float* out);
 *   void power(const float* ri, int rows, int nb, float* out);
 *   void logmel_finish(float* mel, int B, int per_b);
 *   void copy_rows(const float* x, int rows, int D, int ldx, float* y, int ldy);
 */
#ifndef TK_WHISPER_GRAPH_H
#define TK_WHISPER_GRAPH_H

#include <math.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "tk_exact_math.h"
#include "tk_gemm_desc.h"
#include "tk_ggml_blocks.h"

#define TK_WH_NFFT 400
#define TK_WH_HOP 160
#define TK_WH_NBINS 201
#define TK_WH_LN_EPS 1e-5f

struct TkWhisperHP {
    int32_t n_mels, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
    int32_t n_text_ctx, n_text_state, n_text_head, n_text_layer, n_vocab;
    int n_frames() const { return 2 * n_audio_ctx; }
    int n_samples() const { return n_frames() * TK_WH_HOP; }
};

static inline TkWhisperHP tk_whisper_tiny_en() { return TkWhisperHP{80, 1500, 384, 6, 4, 448, 384, 6, 4, 51864}; }

enum TkWhKind { TK_WK_LINEAR_W = 0, TK_WK_BIAS, TK_WK_LN_W, TK_WK_LN_B, TK_WK_EMB, TK_WK_POS, TK_WK_TABLE };

struct TkWhTensor {
    std::string name;
    int64_t rows, cols;
    int kind;
};

/* indices into the tensor list */
struct TkWhLayerIdx {
    int ln1_w, ln1_b, q_w, q_b, k_w, v_w, v_b, o_w, o_b;
    int lnx_w, lnx_b, xq_w, xq_b, xk_w, xv_w, xv_b, xo_w, xo_b; /* decoder only */
    int ln2_w, ln2_b, fc1_w, fc1_b, fc2_w, fc2_b;
};

struct TkWhManifest {
    std::vector<TkWhTensor> t;
    int hann, dft, melw, conv1_w, conv1_b, conv2_w, conv2_b, enc_pos, enc_ln_w, enc_ln_b, tok_emb, dec_pos, dec_ln_w, dec_ln_b;
    std::vector<TkWhLayerIdx> enc, dec;
};

static inline TkWhManifest tk_whisper_manifest(const TkWhisperHP& h) {
    TkWhManifest m;
    auto add = [&](const std::string& n, int64_t r, int64_t c, int kind) { m.t.push_back(TkWhTensor{n, r, c, kind}); return (int)m.t.size() - 1; };
    const int d = h.n_audio_state, dt = h.n_text_state;
    m.hann = add("frontend.hann", 1, TK_WH_NFFT, TK_WK_TABLE);
    m.dft = add("frontend.dft", 2 * TK_WH_NBINS, TK_WH_NFFT, TK_WK_TABLE);
    m.melw = add("frontend.mel_filters", h.n_mels, TK_WH_NBINS, TK_WK_TABLE);
    m.conv1_w = add("encoder.conv1.weight", d, 3 * h.n_mels, TK_WK_LINEAR_W);
    m.conv1_b = add("encoder.conv1.bias", 1, d, TK_WK_BIAS);
    m.conv2_w = add("encoder.conv2.weight", d, 3 * d, TK_WK_LINEAR_W);
    m.conv2_b = add("encoder.conv2.bias", 1, d, TK_WK_BIAS);
    m.enc_pos = add("encoder.positional_embedding", h.n_audio_ctx, d, TK_WK_TABLE);
    auto layer = [&](const std::string& p, int dm, bool cross) {
        TkWhLayerIdx L{};
        L.ln1_w = add(p + "attn_ln.weight", 1, dm, TK_WK_LN_W); L.ln1_b = add(p + "attn_ln.bias", 1, dm, TK_WK_LN_B);
        L.q_w = add(p + "attn.query.weight", dm, dm, TK_WK_LINEAR_W); L.q_b = add(p + "attn.query.bias", 1, dm, TK_WK_BIAS);
        L.k_w = add(p + "attn.key.weight", dm, dm, TK_WK_LINEAR_W);
        L.v_w = add(p + "attn.value.weight", dm, dm, TK_WK_LINEAR_W); L.v_b = add(p + "attn.value.bias", 1, dm, TK_WK_BIAS);
        L.o_w = add(p + "attn.out.weight", dm, dm, TK_WK_LINEAR_W); L.o_b = add(p + "attn.out.bias", 1, dm, TK_WK_BIAS);
        if (cross) {
            L.lnx_w = add(p + "cross_attn_ln.weight", 1, dm, TK_WK_LN_W); L.lnx_b = add(p + "cross_attn_ln.bias", 1, dm, TK_WK_LN_B);
            L.xq_w = add(p + "cross_attn.query.weight", dm, dm, TK_WK_LINEAR_W); L.xq_b = add(p + "cross_attn.query.bias", 1, dm, TK_WK_BIAS);
            L.xk_w = add(p + "cross_attn.key.weight", dm, d, TK_WK_LINEAR_W);
            L.xv_w = add(p + "cross_attn.value.weight", dm, d, TK_WK_LINEAR_W); L.xv_b = add(p + "cross_attn.value.bias", 1, dm, TK_WK_BIAS);
            L.xo_w = add(p + "cross_attn.out.weight", dm, dm, TK_WK_LINEAR_W); L.xo_b = add(p + "cross_attn.out.bias", 1, dm, TK_WK_BIAS);
        }
        L.ln2_w = add(p + "mlp_ln.weight", 1, dm, TK_WK_LN_W); L.ln2_b = add(p + "mlp_ln.bias", 1, dm, TK_WK_LN_B);
        L.fc1_w = add(p + "mlp.0.weight", 4 * dm, dm, TK_WK_LINEAR_W); L.fc1_b = add(p + "mlp.0.bias", 1, 4 * dm, TK_WK_BIAS);
        L.fc2_w = add(p + "mlp.2.weight", dm, 4 * dm, TK_WK_LINEAR_W); L.fc2_b = add(p + "mlp.2.bias", 1, dm, TK_WK_BIAS);
        return L;
    };
    for (int l = 0; l < h.n_audio_layer; ++l) m.enc.push_back(layer("encoder.blocks." + std::to_string(l) + ".", d, false));
    m.enc_ln_w = add("encoder.ln_post.weight", 1, d, TK_WK_LN_W);
    m.enc_ln_b = add("encoder.ln_post.bias", 1, d, TK_WK_LN_B);
    m.tok_emb = add("decoder.token_embedding.weight", h.n_vocab, dt, TK_WK_EMB);
    m.dec_pos = add("decoder.positional_embedding", h.n_text_ctx, dt, TK_WK_POS);
    for (int l = 0; l < h.n_text_layer; ++l) m.dec.push_back(layer("decoder.blocks." + std::to_string(l) + ".", dt, true));
    m.dec_ln_w = add("decoder.ln.weight", 1, dt, TK_WK_LN_W);
    m.dec_ln_b = add("decoder.ln.bias", 1, dt, TK_WK_LN_B);
    return m;
}

/* fixed tables (double precision on the host, rounded once) and seeded synthetic parameters */
static inline void tk_whisper_fill_tensor(const TkWhisperHP& h, const TkWhManifest& m, int idx, uint64_t seed, float* out) {
    const TkWhTensor& t = m.t[idx];
    const double PI = 3.14159265358979323846;
    if (idx == m.hann) {
        for (int n = 0; n < TK_WH_NFFT; ++n) out[n] = (float)(0.5 - 0.5 * cos(2.0 * PI * n / TK_WH_NFFT));
    } else if (idx == m.dft) {
        for (int f = 0; f < TK_WH_NBINS; ++f)
            for (int n = 0; n < TK_WH_NFFT; ++n) {
                const double a = 2.0 * PI * (double)((int64_t)f * n % TK_WH_NFFT) / TK_WH_NFFT;
                out[(size_t)f * TK_WH_NFFT + n] = (float)cos(a);
                out[(size_t)(TK_WH_NBINS + f) * TK_WH_NFFT + n] = (float)sin(a);
            }
    } else if (idx == m.melw) { /* slaney mel scale + slaney area normalisation, 0..8 kHz */
        const int nm = h.n_mels;
        auto hz2mel = [](double f) { const double fsp = 200.0 / 3.0; return f < 1000.0 ? f / fsp : 15.0 + log(f / 1000.0) / (log(6.4) / 27.0); };
        auto mel2hz = [](double mm) { const double fsp = 200.0 / 3.0; return mm < 15.0 ? mm * fsp : 1000.0 * exp((log(6.4) / 27.0) * (mm - 15.0)); };
        std::vector<double> pts(nm + 2);
        const double lo = hz2mel(0.0), hi = hz2mel(8000.0);
        for (int i = 0; i < nm + 2; ++i) pts[i] = mel2hz(lo + (hi - lo) * i / (nm + 1));
        for (int i = 0; i < nm; ++i)
            for (int f = 0; f < TK_WH_NBINS; ++f) {
                const double fr = 8000.0 * f / (TK_WH_NBINS - 1);
                const double lower = (fr - pts[i]) / (pts[i + 1] - pts[i]), upper = (pts[i + 2] - fr) / (pts[i + 2] - pts[i + 1]);
                double w = lower < upper ? lower : upper;
                if (w < 0.0) w = 0.0;
                out[(size_t)i * TK_WH_NBINS + f] = (float)(w * 2.0 / (pts[i + 2] - pts[i]));
            }
    } else if (idx == m.enc_pos) { /* sinusoids(n_audio_ctx, d) */
        const int d = h.n_audio_state, half = d / 2;
        const double inc = log(10000.0) / (half - 1);
        for (int p = 0; p < h.n_audio_ctx; ++p)
            for (int i = 0; i < half; ++i) {
                const double a = p * exp(-inc * i);
                out[(size_t)p * d + i] = (float)sin(a);
                out[(size_t)p * d + half + i] = (float)cos(a);
            }
    } else {
        const int64_t n = t.rows * t.cols;
        float scale = 0.02f, base = 0.0f;
        if (t.kind == TK_WK_LINEAR_W) scale = tk_sqrtf(tk_divf(1.0f, (float)t.cols));
        else if (t.kind == TK_WK_LN_W) { scale = 0.1f; base = 1.0f; }
        else if (t.kind == TK_WK_POS) scale = 0.01f;
        else if (t.kind == TK_WK_EMB) scale = 0.05f;
        for (int64_t i = 0; i < n; ++i) out[i] = base + scale * tk_synth_normal(seed, (uint64_t)(9000 + idx), (uint64_t)i);
    }
}

/* synthetic VAD MLP parameters (the Silero ONNX graph is absent: SURVEY.md §8c) */
static inline void tk_vad_synth(uint64_t seed, int window, int hidden, float* w1, float* b1, float* w2, float* b2) {
    const float s1 = tk_sqrtf(tk_divf(1.0f, (float)window)) * 8.0f, s2 = tk_sqrtf(tk_divf(1.0f, (float)hidden)) * 4.0f;
    for (int64_t i = 0; i < (int64_t)window * hidden; ++i) w1[i] = s1 * tk_synth_normal(seed, 7001, (uint64_t)i);
    for (int i = 0; i < hidden; ++i) b1[i] = 0.1f * tk_synth_normal(seed, 7002, (uint64_t)i);
    for (int i = 0; i < hidden; ++i) w2[i] = s2 * tk_synth_normal(seed, 7003, (uint64_t)i);
    b2[0] = -1.0f;
}

template <class Ops>
struct TkWhisperGraph {
    const TkWhisperHP& h;
    const TkWhManifest& m;
    float* const* W; /* tensor idx -> data */

    static TkGemm lin(const float* A, int M, int K, int lda, const float* Wt, const float* bias, int N, float* C, int ldc, int act = 0,
                      const float* res = nullptr, int ldr = 0) {
        TkGemm g{};
        g.A = A; g.B = Wt; g.C = C; g.bias = bias; g.residual = res;
        g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = K; g.ldc = ldc; g.ldr = ldr;
        g.b_kn = 0; g.act = act; g.alpha = 1.0f; g.batch = 1;
        return g;
    }

    /* pcm: [B][pcm_stride] int16, n_samples valid per row (zero padded to the 30 s window); mel out [B][n_frames][n_mels] */
    float* mel(Ops& o, const int16_t* pcm, int B, int n_samples, int pcm_stride) const {
        const int T = h.n_frames();
        float* fr = o.alloc((size_t)B * T * TK_WH_NFFT);
        o.frames(pcm, B, n_samples, pcm_stride, h.n_samples(), T, W[m.hann], fr);
        float* ri = o.alloc((size_t)B * T * 2 * TK_WH_NBINS);
        o.gemm(lin(fr, B * T, TK_WH_NFFT, TK_WH_NFFT, W[m.dft], nullptr, 2 * TK_WH_NBINS, ri, 2 * TK_WH_NBINS));
        float* pw = o.alloc((size_t)B * T * TK_WH_NBINS);
        o.power(ri, B * T, TK_WH_NBINS, pw);
        float* ml = o.alloc((size_t)B * T * h.n_mels);
        o.gemm(lin(pw, B * T, TK_WH_NBINS, TK_WH_NBINS, W[m.melw], nullptr, h.n_mels, ml, h.n_mels));
        o.logmel_finish(ml, B, T * h.n_mels);
        return ml;
    }

    /* row pitch of the scores scratch: whole 128-byte lines per row (1500 keys -> 1504 floats would do for 16-byte groups; 1536 keeps every
     * row, and every 128-column tile of it, on cache-line boundaries: Q.K^T's stores and the row kernels' 16-byte groups never straddle) */
    static int score_pitch(int Tk) { return (Tk + 31) & ~31; }
    /* attention over `Tk` keys for `Tq` queries per batch element; q/k/v/out row pitch = d; scores scratch [B*nh][Tq][score_pitch(Tk)] */
    void attention(Ops& o, const float* q, const float* k, const float* v, float* out, int B, int Tq, int Tk, int64_t q_bstride, int64_t kv_bstride,
                   int d, int nh, float* scores) const {
        const int hd = d / nh;
        if (Tq == 1 && o.attend1(q, k, v, out, B, Tk, q_bstride, kv_bstride, d, nh)) return; /* decoder steps: one kernel, same arithmetic */
        if (Tq > 1 && o.attend_fused(q, k, v, out, B, Tq, Tk, q_bstride, kv_bstride, d, nh)) return; /* opt-in fast path only */
        const int Tp = score_pitch(Tk);
        TkGemm s{};
        s.A = q; s.B = k; s.C = scores; s.M = Tq; s.N = Tk; s.K = hd; s.lda = d; s.ldb = d; s.ldc = Tp; s.b_kn = 0; s.act = 0;
        s.alpha = tk_divf(1.0f, tk_sqrtf((float)hd));
        s.batch = B * nh; s.batch_inner = nh;
        s.sA = hd; s.sB = hd; s.sC = (int64_t)Tq * Tp;
        s.sA2 = q_bstride; s.sB2 = kv_bstride; s.sC2 = (int64_t)nh * Tq * Tp;
        o.gemm(s);
        o.softmax_rows(scores, B * nh * Tq, Tk, Tp);
        TkGemm p{};
        p.A = scores; p.B = v; p.C = out; p.M = Tq; p.N = hd; p.K = Tk; p.lda = Tp; p.ldb = d; p.ldc = d; p.b_kn = 1; p.act = 0; p.alpha = 1.0f;
        p.batch = B * nh; p.batch_inner = nh;
        p.sA = (int64_t)Tq * Tp; p.sB = hd; p.sC = hd;
        p.sA2 = (int64_t)nh * Tq * Tp; p.sB2 = kv_bstride; p.sC2 = q_bstride;
        o.gemm(p);
    }

    /* mel [B][2*ctx][n_mels] -> encoder states [B][ctx][d] */
    float* encode(Ops& o, const float* ml, int B) const {
        const int T = h.n_frames(), Tc = h.n_audio_ctx, d = h.n_audio_state, nh = h.n_audio_head;
        float* col1 = o.alloc((size_t)B * T * 3 * h.n_mels);
        o.im2col1d(ml, B, T, h.n_mels, h.n_mels, 3, 1, 1, col1);
        float* x1 = o.alloc((size_t)B * T * d);
        o.gemm(lin(col1, B * T, 3 * h.n_mels, 3 * h.n_mels, W[m.conv1_w], W[m.conv1_b], d, x1, d, TK_ACT_GELU));
        float* col2 = o.alloc((size_t)B * Tc * 3 * d);
        o.im2col1d(x1, B, T, d, d, 3, 2, 1, col2);
        float* x = o.alloc((size_t)B * Tc * d);
        o.gemm(lin(col2, B * Tc, 3 * d, 3 * d, W[m.conv2_w], W[m.conv2_b], d, x, d, TK_ACT_GELU));
        o.add_rows(x, W[m.enc_pos], B * Tc, d, Tc);
        float* hb = o.alloc((size_t)B * Tc * d);
        float* q = o.alloc((size_t)B * Tc * d);
        float* k = o.alloc((size_t)B * Tc * d);
        float* v = o.alloc((size_t)B * Tc * d);
        float* at = o.alloc((size_t)B * Tc * d);
        float* ff = o.alloc((size_t)B * Tc * 4 * d);
        float* sc = o.alloc((size_t)B * nh * Tc * score_pitch(Tc));
        const int R = B * Tc;
        for (int l = 0; l < h.n_audio_layer; ++l) {
            const TkWhLayerIdx& L = m.enc[l];
            o.layernorm(x, R, d, W[L.ln1_w], W[L.ln1_b], hb);
            o.gemm(lin(hb, R, d, d, W[L.q_w], W[L.q_b], d, q, d));
            o.gemm(lin(hb, R, d, d, W[L.k_w], nullptr, d, k, d));
            o.gemm(lin(hb, R, d, d, W[L.v_w], W[L.v_b], d, v, d));
            attention(o, q, k, v, at, B, Tc, Tc, (int64_t)Tc * d, (int64_t)Tc * d, d, nh, sc);
            o.gemm(lin(at, R, d, d, W[L.o_w], W[L.o_b], d, x, d, 0, x, d));
            o.layernorm(x, R, d, W[L.ln2_w], W[L.ln2_b], hb);
            { TkGemm g1 = lin(hb, R, d, d, W[L.fc1_w], W[L.fc1_b], 4 * d, ff, 4 * d, TK_ACT_GELU); g1.c_feeds_linear = 1; o.gemm(g1); }
            o.gemm(lin(ff, R, 4 * d, 4 * d, W[L.fc2_w], W[L.fc2_b], d, x, d, 0, x, d));
        }
        float* y = o.alloc((size_t)R * d);
        o.layernorm(x, R, d, W[m.enc_ln_w], W[m.enc_ln_b], y);
        return y;
    }

    struct DecState {
        int B = 0;
        std::vector<float*> xk, xv; /* per layer [B][ctx][dt] */
        std::vector<float*> sk, sv; /* per layer self caches [B][n_text_ctx][dt] */
        float *x, *hb, *q, *k, *v, *at, *ff, *sc, *logits;
        int32_t *tok, *pos, *next;
    };

    DecState begin_decode(Ops& o, const float* enc, int B) const {
        const int Tc = h.n_audio_ctx, d = h.n_audio_state, dt = h.n_text_state;
        DecState s;
        s.B = B;
        for (int l = 0; l < h.n_text_layer; ++l) {
            const TkWhLayerIdx& L = m.dec[l];
            float* xk = o.alloc((size_t)B * Tc * dt);
            float* xv = o.alloc((size_t)B * Tc * dt);
            o.gemm(lin(enc, B * Tc, d, d, W[L.xk_w], nullptr, dt, xk, dt));
            o.gemm(lin(enc, B * Tc, d, d, W[L.xv_w], W[L.xv_b], dt, xv, dt));
            s.xk.push_back(xk); s.xv.push_back(xv);
            s.sk.push_back(o.alloc((size_t)B * h.n_text_ctx * dt));
            s.sv.push_back(o.alloc((size_t)B * h.n_text_ctx * dt));
        }
        s.x = o.alloc((size_t)B * dt); s.hb = o.alloc((size_t)B * dt); s.q = o.alloc((size_t)B * dt);
        s.k = o.alloc((size_t)B * dt); s.v = o.alloc((size_t)B * dt); s.at = o.alloc((size_t)B * dt);
        s.ff = o.alloc((size_t)B * 4 * dt);
        const int tmax = Tc > h.n_text_ctx ? Tc : h.n_text_ctx;
        s.sc = o.alloc((size_t)B * h.n_text_head * score_pitch(tmax));
        s.logits = o.alloc((size_t)B * h.n_vocab);
        s.tok = o.alloc_i32(B); s.pos = o.alloc_i32(B); s.next = o.alloc_i32(B);
        return s;
    }

    /* one decoder step at position p for all B sequences: tokens in s.tok, positions in s.pos (== p); argmax -> s.next */
    void decode_step(Ops& o, DecState& s, int p) const {
        const int B = s.B, Tc = h.n_audio_ctx, dt = h.n_text_state, nh = h.n_text_head, ctx = h.n_text_ctx;
        o.embed_rows(W[m.tok_emb], W[m.dec_pos], s.tok, s.pos, B, dt, s.x);
        for (int l = 0; l < h.n_text_layer; ++l) {
            const TkWhLayerIdx& L = m.dec[l];
            o.layernorm(s.x, B, dt, W[L.ln1_w], W[L.ln1_b], s.hb);
            o.gemm(lin(s.hb, B, dt, dt, W[L.q_w], W[L.q_b], dt, s.q, dt));
            /* k, v of this position go straight into the caches (row p of every sequence) */
            o.gemm(lin(s.hb, B, dt, dt, W[L.k_w], nullptr, dt, s.sk[l] + (size_t)p * dt, ctx * dt));
            o.gemm(lin(s.hb, B, dt, dt, W[L.v_w], W[L.v_b], dt, s.sv[l] + (size_t)p * dt, ctx * dt));
            attention(o, s.q, s.sk[l], s.sv[l], s.at, B, 1, p + 1, dt, (int64_t)ctx * dt, dt, nh, s.sc);
            o.gemm(lin(s.at, B, dt, dt, W[L.o_w], W[L.o_b], dt, s.x, dt, 0, s.x, dt));
            o.layernorm(s.x, B, dt, W[L.lnx_w], W[L.lnx_b], s.hb);
            o.gemm(lin(s.hb, B, dt, dt, W[L.xq_w], W[L.xq_b], dt, s.q, dt));
            attention(o, s.q, s.xk[l], s.xv[l], s.at, B, 1, Tc, dt, (int64_t)Tc * dt, dt, nh, s.sc);
            o.gemm(lin(s.at, B, dt, dt, W[L.xo_w], W[L.xo_b], dt, s.x, dt, 0, s.x, dt));
            o.layernorm(s.x, B, dt, W[L.ln2_w], W[L.ln2_b], s.hb);
            { TkGemm g1 = lin(s.hb, B, dt, dt, W[L.fc1_w], W[L.fc1_b], 4 * dt, s.ff, 4 * dt, TK_ACT_GELU); g1.c_feeds_linear = 1; o.gemm(g1); }
            o.gemm(lin(s.ff, B, 4 * dt, 4 * dt, W[L.fc2_w], W[L.fc2_b], dt, s.x, dt, 0, s.x, dt));
        }
        o.layernorm(s.x, B, dt, W[m.dec_ln_w], W[m.dec_ln_b], s.hb);
        o.gemm(lin(s.hb, B, dt, dt, W[m.tok_emb], nullptr, h.n_vocab, s.logits, h.n_vocab));
        o.argmax_rows(s.logits, B, h.n_vocab, h.n_vocab, s.next);
    }
};

#endif
