/* tk_vad_graph.hip — see tk_vad_graph.h */
#include "tk_vad_graph.h"

#include <stdio.h>
#include <string.h>

#include <algorithm>

#include "../common/tk_exact_math.h"

#define VQ(expr)                                                                                                  \
    do {                                                                                                          \
        hipError_t e__ = (expr);                                                                                  \
        if (e__ != hipSuccess) { error = std::string(#expr) + " failed: " + hipGetErrorString(e__); return false; } \
    } while (0)

/* ------------------------------------------------------------------ kernels (tiny tensors: one thread per output element) ---- */

enum { U_RELU, U_SIGMOID, U_TANH, U_SQRT, U_ABS, U_NEG, U_EXP, U_LOG };
enum { B_ADD, B_SUB, B_MUL, B_DIV, B_POW };

__global__ void k_vg_unary(int op, const float* x, float* y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    float r;
    switch (op) {
        case U_RELU: r = tk_fmaxf(v, 0.0f); break;
        case U_SIGMOID: r = tk_sigmoidf(v); break;
        case U_TANH: r = tk_tanhf(v); break;
        case U_SQRT: r = tk_sqrtf(v); break;
        case U_ABS: r = tk_fabsf(v); break;
        case U_NEG: r = -v; break;
        case U_EXP: r = tk_expf(v); break;
        default: r = tk_logf(v); break;
    }
    y[i] = r;
}

struct VgIdx { int64_t dim[4], sa[4], sb[4]; };

__global__ void k_vg_binary(int op, const float* a, const float* b, float* y, VgIdx ix, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t r = i, oa = 0, ob = 0;
    for (int d = 3; d >= 0; --d) {
        const int64_t c = r % ix.dim[d];
        r /= ix.dim[d];
        oa += c * ix.sa[d];
        ob += c * ix.sb[d];
    }
    const float x = a[oa], z = b[ob];
    float v;
    switch (op) {
        case B_ADD: v = x + z; break;
        case B_SUB: v = x - z; break;
        case B_MUL: v = x * z; break;
        case B_DIV: v = tk_divf(x, z); break;
        default: /* Pow: the exponents these graphs use (magnitude: 2; root: 0.5) */
            v = z == 2.0f ? x * x : z == 0.5f ? tk_sqrtf(x) : z == 1.0f ? x : tk_expf(z * tk_logf(x));
            break;
    }
    y[i] = v;
}

/* generic gather: out element i (coordinates over dim[]) reads x[off + sum c_d * sa[d]] — Transpose, Slice, Concat pieces */
__global__ void k_vg_gather(const float* x, float* y, VgIdx ix, int64_t off, int64_t n, int64_t y_off, VgIdx oy) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t r = i, oa = off, oo = y_off;
    for (int d = 3; d >= 0; --d) {
        const int64_t c = r % ix.dim[d];
        r /= ix.dim[d];
        oa += c * ix.sa[d];
        oo += c * oy.sa[d];
    }
    y[oo] = x[oa];
}

__global__ void k_vg_pad_last(const float* x, float* y, int64_t rows, int64_t L, int64_t pb, int64_t Lo, int reflect, float cval) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * Lo) return;
    const int64_t r = i / Lo;
    int64_t t = i % Lo - pb;
    float v = cval;
    if (t >= 0 && t < L) v = x[r * L + t];
    else if (reflect) {
        if (t < 0) t = -t;
        if (t >= L) t = 2 * (L - 1) - t;
        v = (t >= 0 && t < L) ? x[r * L + t] : cval;
    }
    y[i] = v;
}

/* y[m][t] = (sum_c sum_k x[c][t s - pb + k d] w[m][c][k]) + b[m]: one fma chain, c outer, k inner, ascending */
__global__ void k_vg_conv1d(const float* x, const float* w, const float* b, float* y, int C, int64_t L, int M, int K, int stride, int pb, int dil, int64_t Lo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * Lo) return;
    const int m = (int)(i / Lo);
    const int64_t t = i % Lo;
    float acc = 0.0f;
    for (int c = 0; c < C; ++c)
        for (int k = 0; k < K; ++k) {
            const int64_t p = t * stride - pb + (int64_t)k * dil;
            if (p >= 0 && p < L) acc = tk_fmaf(x[c * L + p], w[((int64_t)m * C + c) * K + k], acc);
        }
    y[i] = b ? acc + b[m] : acc;
}

__global__ void k_vg_reduce_mean(const float* x, float* y, int64_t outer, int64_t axis, int64_t inner) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= outer * inner) return;
    const int64_t o = i / inner, in = i % inner;
    float s = 0.0f;
    for (int64_t a = 0; a < axis; ++a) s = s + x[(o * axis + a) * inner + in];
    y[i] = tk_divf(s, (float)axis);
}

/* ONNX LSTM, forward, batch 1.  X [T][I]; W [4H][I], R [4H][H] (gate rows i, o, f, c); B [8H] = Wb | Rb (or null).  One workgroup of
 * 4H threads: thread g computes gate row g (x part then h part, ascending fma chains, then the two biases), the first H threads update
 * the cell.  y [T][H]; h / c are updated in place (hn / cn). */
__global__ void k_vg_lstm(const float* X, const float* W, const float* R, const float* B, float* h, float* c, float* Y, int T, int I, int H) {
    extern __shared__ float gates[]; /* 4H */
    const int g = threadIdx.x;
    for (int t = 0; t < T; ++t) {
        if (g < 4 * H) {
            float a = 0.0f;
            for (int k = 0; k < I; ++k) a = tk_fmaf(X[(int64_t)t * I + k], W[(int64_t)g * I + k], a);
            for (int k = 0; k < H; ++k) a = tk_fmaf(h[k], R[(int64_t)g * H + k], a);
            if (B) a = (a + B[g]) + B[4 * H + g];
            gates[g] = a;
        }
        __syncthreads();
        if (g < H) {
            const float it = tk_sigmoidf(gates[g]), ot = tk_sigmoidf(gates[H + g]), ft = tk_sigmoidf(gates[2 * H + g]), ct = tk_tanhf(gates[3 * H + g]);
            const float cn = tk_fmaf(ft, c[g], it * ct);
            const float hn = ot * tk_tanhf(cn);
            c[g] = cn;
            h[g] = hn;
            if (Y) Y[(int64_t)t * H + g] = hn;
        }
        __syncthreads();
    }
}

/* ------------------------------------------------------------------ host side ---- */

static dim3 grid_for(int64_t n) { return dim3((unsigned)((n + 127) / 128)); }

TkVadGraph::~TkVadGraph() {
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    for (auto& kv : consts_) if (kv.second.d) (void)hipFree(kv.second.d);
    for (auto& s : states_) if (s.buf) (void)hipFree(s.buf);
    if (arena_) (void)hipFree(arena_);
    if (x_dev_) (void)hipFree(x_dev_);
    if (p_dev_) (void)hipFree(p_dev_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

float* TkVadGraph::alloc(int64_t n) {
    const size_t need = ((size_t)(n > 0 ? n : 1) + 63) & ~(size_t)63;
    if (arena_used_ + need > arena_cap_) return nullptr;
    float* p = arena_ + arena_used_;
    arena_used_ += need;
    return p;
}

static const char* kSupported[] = {"Conv", "Relu", "Sigmoid", "Tanh", "Sqrt", "Abs", "Neg", "Exp", "Log", "Add", "Sub", "Mul", "Div", "Pow", "Slice", "Concat",
                                   "Pad", "Transpose", "ReduceMean", "LSTM", "Unsqueeze", "Squeeze", "Reshape", "Flatten", "Identity", "Cast", "Constant"};

bool TkVadGraph::check_supported(const TkOnnxGraph& g, std::string* err) {
    for (const auto& nd : g.nodes) {
        bool ok = false;
        for (const char* s : kSupported) ok = ok || nd.op == s;
        if (!ok) { *err = "ONNX op '" + nd.op + "' (node '" + nd.name + "') is outside the VAD graph class this path runs"; return false; }
    }
    if (g.outputs.empty()) { *err = "the graph declares no outputs"; return false; }
    bool audio = false;
    for (const auto& vi : g.inputs) audio = audio || vi.elem_type == 1 || vi.elem_type == 0;
    if (!audio) { *err = "the graph has no float input for the audio window"; return false; }
    return true;
}

bool TkVadGraph::load(const char* path, int device, int window, int sample_rate) {
    device_ = device; window_ = window; sample_rate_ = sample_rate;
    if (!g_.load(path)) { error = g_.error; return false; }
    if (!check_supported(g_, &error)) return false;
    /* inputs: the first float input is the audio window; an integer input is the sample rate; the other float inputs are recurrent state */
    std::vector<const TkOnnxValueInfo*> extra;
    for (const auto& vi : g_.inputs) {
        if (vi.elem_type == 7 || vi.elem_type == 6) { sr_in_ = vi.name; continue; }
        if (audio_in_.empty()) { audio_in_ = vi.name; continue; }
        extra.push_back(&vi);
    }
    if (audio_in_.empty()) { error = "the graph has no float input for the audio window"; return false; }
    prob_out_ = g_.outputs[0].name;
    VQ(hipSetDevice(device_));
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess) VQ(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, hi));
    else VQ(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    for (size_t i = 0; i < extra.size(); ++i) {
        State st;
        st.in = extra[i]->name;
        const std::string cand[2] = {st.in + "n", st.in + "N"}; /* h -> hn, c -> cn, state -> stateN */
        for (const auto& o : g_.outputs)
            if (o.name == cand[0] || o.name == cand[1]) st.out = o.name;
        if (st.out.empty() && i + 1 < g_.outputs.size()) st.out = g_.outputs[i + 1].name;
        if (st.out.empty()) { error = "recurrent input '" + st.in + "' has no matching output"; return false; }
        int64_t n = 1;
        for (int64_t d : extra[i]->dims) { const int64_t dd = d > 0 ? d : 1; st.shape.push_back(dd); n *= dd; } /* symbolic batch = 1 */
        if (st.shape.empty() || n > (1 << 20)) { error = "recurrent input '" + st.in + "' needs a static shape"; return false; }
        VQ(hipMalloc((void**)&st.buf, (size_t)n * 4));
        VQ(hipMemset(st.buf, 0, (size_t)n * 4));
        states_.push_back(st);
    }
    for (const auto& kv : g_.init) {
        Val v;
        v.shape = kv.second.dims;
        if (!kv.second.f.empty()) {
            VQ(hipMalloc((void**)&v.d, kv.second.f.size() * 4));
            VQ(hipMemcpy(v.d, kv.second.f.data(), kv.second.f.size() * 4, hipMemcpyHostToDevice));
        } else if (!kv.second.i.empty() || kv.second.count() == 0) {
            v.is_int = true;
            v.ints = kv.second.i;
        } else { error = "initialiser '" + kv.first + "' has a data type this path does not read"; return false; }
        consts_[kv.first] = v;
    }
    for (const auto& nd : g_.nodes) {
        if (nd.op != "Constant" || nd.out.empty()) continue;
        auto it = nd.attr.find("value");
        if (it == nd.attr.end() || !it->second.has_t) { error = "Constant node without a tensor value"; return false; }
        const TkOnnxTensor& t = it->second.t;
        Val v;
        v.shape = t.dims;
        if (!t.f.empty()) {
            VQ(hipMalloc((void**)&v.d, t.f.size() * 4));
            VQ(hipMemcpy(v.d, t.f.data(), t.f.size() * 4, hipMemcpyHostToDevice));
        } else { v.is_int = true; v.ints = t.i; }
        consts_[nd.out[0]] = v;
    }
    arena_cap_ = (size_t)1 << 22; /* 16 MiB of activations: far more than a 30 ms window needs in any graph of this class */
    VQ(hipMalloc((void**)&arena_, arena_cap_ * 4));
    VQ(hipMalloc((void**)&p_dev_, 4));
    /* one dry run finds shape or broadcasting errors at load time, as session creation does in the reference */
    std::vector<float> zero((size_t)window_, 0.0f);
    float p = 0.0f;
    if (!infer(zero.data(), 1, &p)) return false;
    return reset();
}

bool TkVadGraph::reset() {
    VQ(hipSetDevice(device_));
    for (auto& s : states_) {
        int64_t n = 1;
        for (int64_t d : s.shape) n *= d;
        VQ(hipMemsetAsync(s.buf, 0, (size_t)n * 4, stream_));
    }
    VQ(hipStreamSynchronize(stream_));
    return true;
}

bool TkVadGraph::infer(const float* windows_host, int n, float* prob_host) {
    if (n <= 0) return true;
    VQ(hipSetDevice(device_));
    if (n > xcap_) {
        if (x_dev_) (void)hipFree(x_dev_);
        if (p_dev_) (void)hipFree(p_dev_);
        x_dev_ = p_dev_ = nullptr;
        const int want = n < 128 ? 128 : n;
        VQ(hipMalloc((void**)&x_dev_, (size_t)want * window_ * 4));
        VQ(hipMalloc((void**)&p_dev_, (size_t)want * 4));
        xcap_ = want;
    }
    VQ(hipMemcpyAsync(x_dev_, windows_host, (size_t)n * window_ * 4, hipMemcpyHostToDevice, stream_));
    for (int i = 0; i < n; ++i)
        if (!run_window(x_dev_ + (size_t)i * window_, p_dev_ + i)) return false;
    VQ(hipGetLastError());
    VQ(hipMemcpyAsync(prob_host, p_dev_, (size_t)n * 4, hipMemcpyDeviceToHost, stream_));
    VQ(hipStreamSynchronize(stream_));
    return true;
}

bool TkVadGraph::run_window(const float* x_dev, float* prob_dev) {
    arena_used_ = 0;
    std::map<std::string, Val> v = consts_;
    Val in;
    in.d = const_cast<float*>(x_dev);
    in.shape = {1, window_};
    v[audio_in_] = in;
    if (!sr_in_.empty()) { Val sr; sr.is_int = true; sr.shape = {1}; sr.ints = {sample_rate_}; v[sr_in_] = sr; }
    for (auto& s : states_) { Val sv; sv.d = s.buf; sv.shape = s.shape; v[s.in] = sv; }
    for (const auto& nd : g_.nodes) {
        if (nd.op == "Constant") continue;
        if (!exec(nd, v)) { if (error.find(nd.op) == std::string::npos) error = nd.op + " (node '" + nd.name + "'): " + error; return false; }
    }
    auto po = v.find(prob_out_);
    if (po == v.end() || !po->second.d || po->second.count() < 1) { error = "the graph did not produce its first output"; return false; }
    VQ(hipMemcpyAsync(prob_dev, po->second.d, 4, hipMemcpyDeviceToDevice, stream_));
    for (auto& s : states_) { /* the next window starts from this window's final state */
        auto so = v.find(s.out);
        int64_t n = 1;
        for (int64_t d : s.shape) n *= d;
        if (so == v.end() || !so->second.d || so->second.count() != n) { error = "state output '" + s.out + "' is missing or has the wrong size"; return false; }
        if (so->second.d != s.buf) VQ(hipMemcpyAsync(s.buf, so->second.d, (size_t)n * 4, hipMemcpyDeviceToDevice, stream_));
    }
    return true;
}

static std::vector<int64_t> strides_of(const std::vector<int64_t>& sh) {
    std::vector<int64_t> st(sh.size(), 1);
    for (int i = (int)sh.size() - 2; i >= 0; --i) st[(size_t)i] = st[(size_t)i + 1] * sh[(size_t)i + 1];
    return st;
}

bool TkVadGraph::exec(const TkOnnxNode& nd, std::map<std::string, Val>& v) {
    auto in = [&](size_t i) -> Val* {
        if (i >= nd.in.size() || nd.in[i].empty()) return nullptr;
        auto it = v.find(nd.in[i]);
        return it == v.end() ? nullptr : &it->second;
    };
    auto need = [&](size_t i, bool want_int) -> Val* {
        Val* x = in(i);
        if (!x) { error = "input " + std::to_string(i) + " is missing"; return nullptr; }
        if (x->is_int != want_int) { error = "input " + std::to_string(i) + (want_int ? " must be an integer tensor" : " must be a float tensor"); return nullptr; }
        return x;
    };
    auto out_f = [&](size_t i, const std::vector<int64_t>& shape) -> Val* {
        Val o;
        o.shape = shape;
        o.d = alloc(o.count());
        if (!o.d) { error = "activation arena exhausted"; return nullptr; }
        v[nd.out[i]] = o;
        return &v[nd.out[i]];
    };
    auto ints_arg = [&](const char* attr, size_t input_idx, std::vector<int64_t>* dst) { /* attribute (old opsets) or integer input (new ones) */
        if (const std::vector<int64_t>* a = nd.aints(attr)) { *dst = *a; return true; }
        Val* x = in(input_idx);
        if (x && x->is_int) { *dst = x->ints; return true; }
        return false;
    };
    const std::string& op = nd.op;
    if (nd.out.empty()) { error = "node without outputs"; return false; }

    int uop = op == "Relu" ? U_RELU : op == "Sigmoid" ? U_SIGMOID : op == "Tanh" ? U_TANH : op == "Sqrt" ? U_SQRT : op == "Abs" ? U_ABS : op == "Neg" ? U_NEG
              : op == "Exp" ? U_EXP : op == "Log" ? U_LOG : -1;
    if (uop >= 0) {
        Val* x = need(0, false);
        if (!x) return false;
        const Val xc = *x;
        Val* y = out_f(0, xc.shape);
        if (!y) return false;
        hipLaunchKernelGGL(k_vg_unary, grid_for(xc.count()), dim3(128), 0, stream_, uop, xc.d, y->d, xc.count());
        return true;
    }
    int bop = op == "Add" ? B_ADD : op == "Sub" ? B_SUB : op == "Mul" ? B_MUL : op == "Div" ? B_DIV : op == "Pow" ? B_POW : -1;
    if (bop >= 0) {
        Val* a = need(0, false);
        Val* b = a ? need(1, false) : nullptr;
        if (!a || !b) return false;
        const Val ac = *a, bc = *b;
        const size_t r = std::max(ac.shape.size(), bc.shape.size());
        if (r > 4) { error = "rank > 4"; return false; }
        std::vector<int64_t> as(r, 1), bs(r, 1), os(r, 1);
        for (size_t i = 0; i < ac.shape.size(); ++i) as[r - ac.shape.size() + i] = ac.shape[i];
        for (size_t i = 0; i < bc.shape.size(); ++i) bs[r - bc.shape.size() + i] = bc.shape[i];
        for (size_t i = 0; i < r; ++i) {
            if (as[i] != bs[i] && as[i] != 1 && bs[i] != 1) { error = "shapes do not broadcast"; return false; }
            os[i] = std::max(as[i], bs[i]);
        }
        const std::vector<int64_t> sa = strides_of(as), sb = strides_of(bs);
        VgIdx ix{};
        for (int d = 0; d < 4; ++d) { ix.dim[d] = 1; ix.sa[d] = ix.sb[d] = 0; }
        for (size_t i = 0; i < r; ++i) {
            const size_t d = 4 - r + i;
            ix.dim[d] = os[i];
            ix.sa[d] = as[i] == 1 ? 0 : sa[i];
            ix.sb[d] = bs[i] == 1 ? 0 : sb[i];
        }
        Val* y = out_f(0, os);
        if (!y) return false;
        hipLaunchKernelGGL(k_vg_binary, grid_for(y->count()), dim3(128), 0, stream_, bop, ac.d, bc.d, y->d, ix, y->count());
        return true;
    }
    if (op == "Identity" || op == "Cast" || op == "Unsqueeze" || op == "Squeeze" || op == "Reshape" || op == "Flatten") {
        Val* x = in(0);
        if (!x) { error = "input 0 is missing"; return false; }
        Val y = *x;
        const int64_t n = y.count();
        if (op == "Unsqueeze") {
            std::vector<int64_t> axes;
            if (!ints_arg("axes", 1, &axes)) { error = "axes are missing"; return false; }
            const int64_t r = (int64_t)y.shape.size() + (int64_t)axes.size();
            for (auto& a : axes) if (a < 0) a += r;
            std::sort(axes.begin(), axes.end());
            for (int64_t a : axes) { if (a < 0 || a > (int64_t)y.shape.size()) { error = "bad axis"; return false; } y.shape.insert(y.shape.begin() + a, 1); }
        } else if (op == "Squeeze") {
            std::vector<int64_t> axes;
            std::vector<int64_t> ns;
            const bool have = ints_arg("axes", 1, &axes);
            for (auto& a : axes) if (a < 0) a += (int64_t)y.shape.size();
            for (size_t i = 0; i < y.shape.size(); ++i) {
                const bool drop = have ? std::find(axes.begin(), axes.end(), (int64_t)i) != axes.end() : y.shape[i] == 1;
                if (drop && y.shape[i] != 1) { error = "squeezed dimension is not 1"; return false; }
                if (!drop) ns.push_back(y.shape[i]);
            }
            y.shape = ns;
        } else if (op == "Reshape") {
            Val* s = need(1, true);
            if (!s) return false;
            std::vector<int64_t> ns = s->ints;
            int64_t known = 1, neg = -1;
            for (size_t i = 0; i < ns.size(); ++i) {
                if (ns[i] == 0) { if (i >= y.shape.size()) { error = "bad 0 in shape"; return false; } ns[i] = y.shape[i]; }
                if (ns[i] == -1) neg = (int64_t)i; else known *= ns[i];
            }
            if (neg >= 0) { if (known == 0 || n % known) { error = "cannot infer -1"; return false; } ns[(size_t)neg] = n / known; known *= ns[(size_t)neg]; }
            if (known != n) { error = "element count changes"; return false; }
            y.shape = ns;
        } else if (op == "Flatten") {
            int64_t ax = nd.ai("axis", 1);
            if (ax < 0) ax += (int64_t)y.shape.size();
            int64_t a = 1, b = 1;
            for (size_t i = 0; i < y.shape.size(); ++i) ((int64_t)i < ax ? a : b) *= y.shape[i];
            y.shape = {a, b};
        } else if (op == "Cast" && nd.ai("to", 1) != 1 && !y.is_int) { error = "only casts to float are supported on float tensors"; return false; }
        v[nd.out[0]] = y;
        return true;
    }
    if (op == "Transpose") {
        Val* x = need(0, false);
        if (!x) return false;
        const Val xc = *x;
        const size_t r = xc.shape.size();
        if (r > 4) { error = "rank > 4"; return false; }
        std::vector<int64_t> perm;
        if (const auto* p = nd.aints("perm")) perm = *p;
        else for (size_t i = 0; i < r; ++i) perm.push_back((int64_t)(r - 1 - i));
        if (perm.size() != r) { error = "perm has the wrong length"; return false; }
        const std::vector<int64_t> sx = strides_of(xc.shape);
        std::vector<int64_t> os(r);
        VgIdx ix{}, oy{};
        for (int d = 0; d < 4; ++d) { ix.dim[d] = 1; ix.sa[d] = 0; oy.sa[d] = 0; }
        for (size_t i = 0; i < r; ++i) { if (perm[i] < 0 || perm[i] >= (int64_t)r) { error = "bad perm"; return false; } os[i] = xc.shape[(size_t)perm[i]]; }
        const std::vector<int64_t> so = strides_of(os);
        for (size_t i = 0; i < r; ++i) { const size_t d = 4 - r + i; ix.dim[d] = os[i]; ix.sa[d] = sx[(size_t)perm[i]]; oy.sa[d] = so[i]; }
        Val* y = out_f(0, os);
        if (!y) return false;
        hipLaunchKernelGGL(k_vg_gather, grid_for(y->count()), dim3(128), 0, stream_, xc.d, y->d, ix, (int64_t)0, y->count(), (int64_t)0, oy);
        return true;
    }
    if (op == "Slice") {
        Val* x = need(0, false);
        if (!x) return false;
        const Val xc = *x;
        const size_t r = xc.shape.size();
        if (r > 4) { error = "rank > 4"; return false; }
        std::vector<int64_t> starts, ends, axes, steps;
        if (!ints_arg("starts", 1, &starts) || !ints_arg("ends", 2, &ends)) { error = "starts / ends are missing"; return false; }
        if (!ints_arg("axes", 3, &axes)) for (size_t i = 0; i < starts.size(); ++i) axes.push_back((int64_t)i);
        if (Val* st = in(4)) { if (st->is_int) steps = st->ints; }
        std::vector<int64_t> os = xc.shape, begin(r, 0);
        for (size_t i = 0; i < axes.size(); ++i) {
            int64_t a = axes[i] < 0 ? axes[i] + (int64_t)r : axes[i];
            if (a < 0 || a >= (int64_t)r || i >= starts.size() || i >= ends.size()) { error = "bad axes"; return false; }
            if (i < steps.size() && steps[i] != 1) { error = "only step 1 is supported"; return false; }
            const int64_t dim = xc.shape[(size_t)a];
            int64_t s = starts[i] < 0 ? starts[i] + dim : starts[i], e = ends[i] < 0 ? ends[i] + dim : ends[i];
            s = std::min(std::max<int64_t>(s, 0), dim);
            e = std::min(std::max<int64_t>(e, 0), dim);
            begin[(size_t)a] = s;
            os[(size_t)a] = e > s ? e - s : 0;
        }
        const std::vector<int64_t> sx = strides_of(xc.shape), so = strides_of(os);
        VgIdx ix{}, oy{};
        for (int d = 0; d < 4; ++d) { ix.dim[d] = 1; ix.sa[d] = 0; oy.sa[d] = 0; }
        int64_t off = 0;
        for (size_t i = 0; i < r; ++i) { const size_t d = 4 - r + i; ix.dim[d] = os[i]; ix.sa[d] = sx[i]; oy.sa[d] = so[i]; off += begin[i] * sx[i]; }
        Val* y = out_f(0, os);
        if (!y) return false;
        if (y->count() > 0) hipLaunchKernelGGL(k_vg_gather, grid_for(y->count()), dim3(128), 0, stream_, xc.d, y->d, ix, off, y->count(), (int64_t)0, oy);
        return true;
    }
    if (op == "Concat") {
        std::vector<Val> parts;
        for (size_t i = 0; i < nd.in.size(); ++i) { Val* x = need(i, false); if (!x) return false; parts.push_back(*x); }
        if (parts.empty()) { error = "no inputs"; return false; }
        const size_t r = parts[0].shape.size();
        if (r > 4) { error = "rank > 4"; return false; }
        int64_t ax = nd.ai("axis", 0);
        if (ax < 0) ax += (int64_t)r;
        std::vector<int64_t> os = parts[0].shape;
        os[(size_t)ax] = 0;
        for (const Val& p : parts) {
            if (p.shape.size() != r) { error = "ranks differ"; return false; }
            for (size_t i = 0; i < r; ++i) if ((int64_t)i != ax && p.shape[i] != parts[0].shape[i]) { error = "shapes differ off the axis"; return false; }
            os[(size_t)ax] += p.shape[(size_t)ax];
        }
        Val* y = out_f(0, os);
        if (!y) return false;
        const std::vector<int64_t> so = strides_of(os);
        int64_t at = 0;
        for (const Val& p : parts) {
            const std::vector<int64_t> sp = strides_of(p.shape);
            VgIdx ix{}, oy{};
            for (int d = 0; d < 4; ++d) { ix.dim[d] = 1; ix.sa[d] = 0; oy.sa[d] = 0; }
            for (size_t i = 0; i < r; ++i) { const size_t d = 4 - r + i; ix.dim[d] = p.shape[i]; ix.sa[d] = sp[i]; oy.sa[d] = so[i]; }
            if (p.count() > 0) hipLaunchKernelGGL(k_vg_gather, grid_for(p.count()), dim3(128), 0, stream_, p.d, y->d, ix, (int64_t)0, p.count(), at * so[(size_t)ax], oy);
            at += p.shape[(size_t)ax];
        }
        return true;
    }
    if (op == "Pad") {
        Val* x = need(0, false);
        if (!x) return false;
        const Val xc = *x;
        std::vector<int64_t> pads;
        if (!ints_arg("pads", 1, &pads)) { error = "pads are missing"; return false; }
        const size_t r = xc.shape.size();
        if (pads.size() != 2 * r) { error = "pads have the wrong length"; return false; }
        for (size_t i = 0; i + 1 < r; ++i) if (pads[i] != 0 || pads[r + i] != 0) { error = "only the last axis can be padded"; return false; }
        const std::string mode = nd.as("mode", "constant");
        if (mode != "constant" && mode != "reflect") { error = "pad mode '" + mode + "' is not supported"; return false; }
        const int64_t L = xc.shape[r - 1], pb = pads[r - 1], pe = pads[2 * r - 1], Lo = L + pb + pe;
        if (pb < 0 || pe < 0 || (mode == "reflect" && (pb >= L || pe >= L))) { error = "bad pad amounts"; return false; }
        std::vector<int64_t> os = xc.shape;
        os[r - 1] = Lo;
        Val* y = out_f(0, os);
        if (!y) return false;
        const int64_t rows = xc.count() / (L > 0 ? L : 1);
        hipLaunchKernelGGL(k_vg_pad_last, grid_for(y->count()), dim3(128), 0, stream_, xc.d, y->d, rows, L, pb, Lo, mode == "reflect" ? 1 : 0, 0.0f);
        return true;
    }
    if (op == "Conv") {
        Val* x = need(0, false);
        Val* w = x ? need(1, false) : nullptr;
        if (!x || !w) return false;
        const Val xc = *x, wc = *w;
        Val* b = in(2);
        if (xc.shape.size() != 3 || wc.shape.size() != 3 || xc.shape[0] != 1) { error = "only 1-D convolutions over [1, C, L] are supported"; return false; }
        if (nd.ai("group", 1) != 1) { error = "grouped convolutions are not supported"; return false; }
        const int C = (int)xc.shape[1], M = (int)wc.shape[0], K = (int)wc.shape[2];
        const int64_t L = xc.shape[2];
        if (wc.shape[1] != C) { error = "channel counts differ"; return false; }
        int stride = 1, dil = 1, pb = 0, pe = 0;
        if (const auto* s = nd.aints("strides")) if (!s->empty()) stride = (int)(*s)[0];
        if (const auto* d = nd.aints("dilations")) if (!d->empty()) dil = (int)(*d)[0];
        if (const auto* p = nd.aints("pads")) if (p->size() == 2) { pb = (int)(*p)[0]; pe = (int)(*p)[1]; }
        const std::string ap = nd.as("auto_pad", "NOTSET");
        if (ap != "NOTSET" && ap != "VALID") { error = "auto_pad '" + ap + "' is not supported"; return false; }
        const int64_t Lo = (L + pb + pe - (int64_t)dil * (K - 1) - 1) / stride + 1;
        if (stride < 1 || Lo < 1) { error = "empty output"; return false; }
        if (b && (b->is_int || b->count() != M)) { error = "bias has the wrong size"; return false; }
        const float* bd = b ? b->d : nullptr;
        Val* y = out_f(0, {1, M, Lo});
        if (!y) return false;
        hipLaunchKernelGGL(k_vg_conv1d, grid_for((int64_t)M * Lo), dim3(128), 0, stream_, xc.d, wc.d, bd, y->d, C, L, M, K, stride, pb, dil, Lo);
        return true;
    }
    if (op == "ReduceMean") {
        Val* x = need(0, false);
        if (!x) return false;
        const Val xc = *x;
        std::vector<int64_t> axes;
        if (!ints_arg("axes", 1, &axes)) for (size_t i = 0; i < xc.shape.size(); ++i) axes.push_back((int64_t)i);
        for (auto& a : axes) if (a < 0) a += (int64_t)xc.shape.size();
        std::sort(axes.begin(), axes.end());
        for (size_t i = 1; i < axes.size(); ++i) if (axes[i] != axes[i - 1] + 1) { error = "reduced axes must be adjacent"; return false; }
        if (axes.empty() || axes.back() >= (int64_t)xc.shape.size()) { error = "bad axes"; return false; }
        int64_t outer = 1, mid = 1, inner = 1;
        std::vector<int64_t> os;
        const bool keep = nd.ai("keepdims", 1) != 0;
        for (size_t i = 0; i < xc.shape.size(); ++i) {
            if ((int64_t)i < axes.front()) { outer *= xc.shape[i]; os.push_back(xc.shape[i]); }
            else if ((int64_t)i <= axes.back()) { mid *= xc.shape[i]; if (keep) os.push_back(1); }
            else { inner *= xc.shape[i]; os.push_back(xc.shape[i]); }
        }
        Val* y = out_f(0, os);
        if (!y) return false;
        hipLaunchKernelGGL(k_vg_reduce_mean, grid_for(outer * inner), dim3(128), 0, stream_, xc.d, y->d, outer, mid, inner);
        return true;
    }
    if (op == "LSTM") {
        Val* X = need(0, false);
        Val* W = X ? need(1, false) : nullptr;
        Val* R = W ? need(2, false) : nullptr;
        if (!X || !W || !R) return false;
        const Val xc = *X, wc = *W, rc = *R;
        if (nd.as("direction", "forward") != "forward") { error = "only the forward direction is supported"; return false; }
        if (xc.shape.size() != 3 || xc.shape[1] != 1 || wc.shape.size() != 3 || wc.shape[0] != 1 || rc.shape.size() != 3) { error = "expects X [T, 1, I], W [1, 4H, I], R [1, 4H, H]"; return false; }
        const int T = (int)xc.shape[0], I = (int)xc.shape[2], H = (int)(wc.shape[1] / 4);
        if (nd.ai("hidden_size", H) != H || wc.shape[2] != I || rc.shape[1] != 4 * H || rc.shape[2] != H || 4 * H > 1024) { error = "inconsistent LSTM geometry (hidden size up to 256)"; return false; }
        Val* B = in(3);
        if (B && (B->is_int || B->count() != 8 * H)) { error = "B must hold 8H values"; return false; }
        const float* bd = B ? B->d : nullptr;
        Val* h0 = in(5);
        Val* c0 = in(6);
        if ((h0 && (h0->is_int || h0->count() != H)) || (c0 && (c0->is_int || c0->count() != H))) { error = "initial state must hold H values"; return false; }
        const Val h0c = h0 ? *h0 : Val(), c0c = c0 ? *c0 : Val();
        /* outputs: Y [T, 1, 1, H], Y_h [1, 1, H], Y_c [1, 1, H]; the running state lives in Y_h / Y_c */
        float* hbuf = alloc(H);
        float* cbuf = alloc(H);
        float* ybuf = alloc((int64_t)T * H);
        if (!hbuf || !cbuf || !ybuf) { error = "activation arena exhausted"; return false; }
        if (h0c.d) VQ(hipMemcpyAsync(hbuf, h0c.d, (size_t)H * 4, hipMemcpyDeviceToDevice, stream_)); else VQ(hipMemsetAsync(hbuf, 0, (size_t)H * 4, stream_));
        if (c0c.d) VQ(hipMemcpyAsync(cbuf, c0c.d, (size_t)H * 4, hipMemcpyDeviceToDevice, stream_)); else VQ(hipMemsetAsync(cbuf, 0, (size_t)H * 4, stream_));
        hipLaunchKernelGGL(k_vg_lstm, dim3(1), dim3(4 * H), (size_t)4 * H * sizeof(float), stream_, xc.d, wc.d, rc.d, bd, hbuf, cbuf, ybuf, T, I, H);
        Val y; y.d = ybuf; y.shape = {T, 1, 1, H};
        Val yh; yh.d = hbuf; yh.shape = {1, 1, H};
        Val yc; yc.d = cbuf; yc.shape = {1, 1, H};
        if (nd.out.size() > 0 && !nd.out[0].empty()) v[nd.out[0]] = y;
        if (nd.out.size() > 1 && !nd.out[1].empty()) v[nd.out[1]] = yh;
        if (nd.out.size() > 2 && !nd.out[2].empty()) v[nd.out[2]] = yc;
        return true;
    }
    error = "unsupported op";
    return false;
}
