/*
 * tk_onnx_exec_seq.hip — the token-sequence half of the graph executor: what a DPT / Swin-transformer depth export adds to the
 * convolutional image graphs of tk_onnx_exec.hip.  The reference names that model class for its depth stream
 * (src/vision/tk_depth_midas.c:8 "DPT-SwinV2-Tiny-256", tests/tk_cortex_test.cpp:42 dpt-swin-tiny.onnx) and runs it inside ONNX Runtime
 * (tk_depth_midas.c:397-440); here every node is one or a few HIP kernels on the owner's stream:
 *   LayerNormalization (last axis), Erf, Gelu (exact and tanh forms), batched MatMul of two activations (QK^T, PV: the exact fp32 MFMA GEMM
 *   of tk_nn_kernels with batch strides), Gather (float data by constant / computed integer indices: relative-position-bias tables, the
 *   q / k / v split of a packed projection), ReduceSum / ReduceL2 / ReduceMax / ReduceMin (cosine attention's row norms), Expand, Max / Min,
 *   Where over a constant mask, ConvTranspose (the DPT reassemble stage's learned upsampling), Shape and the integer arithmetic of shape
 *   sub-graphs (Gather / Concat / Add / Sub / Mul / Div / Cast / Unsqueeze on host integer tensors), Slice with a positive step (patch merging's
 *   x[:, 0::2, 0::2, :]); rank-6 Transpose / Reshape for window partition and merge live in tk_onnx_exec.hip.
 * Arithmetic: fp32 throughout; contractions are k-ascending fma chains; erf is the device library's (<= 1 ulp); results are compared with
 * torch at 2e-5 of the tensor's scale (tests/test_depth_gpu.py), not bit for bit — there is no CPU engine on the reference side to match.
 */
#include "tk_onnx_exec.h"

#include <math.h>
#include <string.h>

#include <algorithm>

#include "../common/tk_exact_math.h"
#include "tk_nn_kernels.h"

#define SQ(expr)                                                                                                 \
    do {                                                                                                         \
        hipError_t e__ = (expr);                                                                                 \
        if (e__ != hipSuccess) { error = std::string(#expr) + ": " + hipGetErrorString(e__); return false; }     \
    } while (0)

namespace {

constexpr int RANK = 6;
struct SqIdx { int64_t dim[RANK], sa[RANK], sb[RANK], sc[RANK]; };

dim3 grid_of(int64_t n) { return dim3((unsigned)((n + 127) / 128)); }

std::vector<int64_t> strides(const std::vector<int64_t>& sh) {
    std::vector<int64_t> st(sh.size(), 1);
    for (int i = (int)sh.size() - 2; i >= 0; --i) st[(size_t)i] = st[(size_t)i + 1] * sh[(size_t)i + 1];
    return st;
}

enum { E_ERF, E_GELU, E_GELU_TANH };
__global__ void k_sq_erf(int op, const float* x, float* y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    float r;
    if (op == E_ERF) r = erff(v);
    else if (op == E_GELU) r = (0.5f * v) * (1.0f + erff(v * 0.70710678118654752440f));
    else r = tk_geluf(v);
    y[i] = r;
}

enum { R_SUM, R_L2, R_MAX, R_MIN, R_MEAN };
__global__ void k_sq_reduce(int op, const float* x, float* y, int64_t outer, int64_t axis, int64_t inner) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= outer * inner) return;
    const int64_t o = i / inner, in = i % inner;
    const float* p = x + o * axis * inner + in;
    float s = (op == R_MAX || op == R_MIN) ? p[0] : 0.0f;
    for (int64_t a = 0; a < axis; ++a) {
        const float v = p[a * inner];
        if (op == R_L2) s = tk_fmaf(v, v, s);
        else if (op == R_MAX) s = tk_fmaxf(s, v);
        else if (op == R_MIN) s = v < s ? v : s;
        else s = s + v;
    }
    if (op == R_L2) s = tk_sqrtf(s);
    if (op == R_MEAN) s = tk_divf(s, (float)axis);
    y[i] = s;
}

/* y[i] = f(a[...], b[...], c[...]) over a broadcast index space: op 0 max(a, b), 1 min(a, b), 2 a (Expand), 3 c != 0 ? a : b (Where) */
__global__ void k_sq_bcast(int op, const float* a, const float* b, const float* c, float* y, SqIdx ix, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t r = i, oa = 0, ob = 0, oc = 0;
    for (int d = RANK - 1; d >= 0; --d) {
        const int64_t q = r % ix.dim[d];
        r /= ix.dim[d];
        oa += q * ix.sa[d];
        ob += q * ix.sb[d];
        oc += q * ix.sc[d];
    }
    float v;
    if (op == 0) v = tk_fmaxf(a[oa], b[ob]);
    else if (op == 1) { const float x = a[oa], z = b[ob]; v = x < z ? x : z; }
    else if (op == 2) v = a[oa];
    else v = c[oc] != 0.0f ? a[oa] : b[ob];
    y[i] = v;
}

/* y[outer][j][inner] = x[outer][idx[j]][inner] */
__global__ void k_sq_gather(const float* x, const int32_t* idx, float* y, int64_t outer, int64_t axis, int64_t nidx, int64_t inner) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= outer * nidx * inner) return;
    const int64_t in = i % inner, j = (i / inner) % nidx, o = i / (inner * nidx);
    y[i] = x[(o * axis + idx[j]) * inner + in];
}

/* strided gather (Slice with steps): output coordinates over dim[], source offset off + sum c_d * sa[d] */
__global__ void k_sq_strided(const float* x, float* y, SqIdx ix, int64_t off, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t r = i, oa = off;
    for (int d = RANK - 1; d >= 0; --d) {
        oa += (r % ix.dim[d]) * ix.sa[d];
        r /= ix.dim[d];
    }
    y[i] = x[oa];
}

/* ONNX ConvTranspose, groups 1, dilation 1: y[n][co][oy][ox] = b[co] + sum_ci sum_ky sum_kx x[n][ci][iy][ix] w[ci][co][ky][kx] with
 * oy = iy * sh - pt + ky (one fma chain: ci outer, ky, kx inner, ascending; the bias enters last) */
struct CtParams { int N, Ci, H, W, Co, kh, kw, sh, sw, pt, pl, Ho, Wo; };
__global__ void k_sq_conv_transpose(const float* x, const float* w, const float* b, float* y, CtParams p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ox = (int)(i % p.Wo), oy = (int)((i / p.Wo) % p.Ho), co = (int)((i / ((int64_t)p.Wo * p.Ho)) % p.Co), nb = (int)(i / ((int64_t)p.Wo * p.Ho * p.Co));
    float acc = 0.0f;
    for (int ci = 0; ci < p.Ci; ++ci)
        for (int ky = 0; ky < p.kh; ++ky) {
            const int ty = oy + p.pt - ky;
            if (ty < 0 || ty % p.sh) continue;
            const int iy = ty / p.sh;
            if (iy >= p.H) continue;
            for (int kx = 0; kx < p.kw; ++kx) {
                const int tx = ox + p.pl - kx;
                if (tx < 0 || tx % p.sw) continue;
                const int jx = tx / p.sw;
                if (jx >= p.W) continue;
                acc = tk_fmaf(x[(((int64_t)nb * p.Ci + ci) * p.H + iy) * p.W + jx], w[(((int64_t)ci * p.Co + co) * p.kh + ky) * p.kw + kx], acc);
            }
        }
    y[i] = b ? acc + b[co] : acc;
}

__global__ void k_sq_fill(float* y, float v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = v;
}

bool broadcast_shape(const std::vector<std::vector<int64_t>>& ins, std::vector<int64_t>* os) {
    size_t r = 0;
    for (const auto& s : ins) r = std::max(r, s.size());
    os->assign(r, 1);
    for (const auto& s : ins)
        for (size_t i = 0; i < s.size(); ++i) {
            int64_t& o = (*os)[r - s.size() + i];
            if (s[i] != 1) { if (o != 1 && o != s[i]) return false; o = s[i]; }
        }
    return true;
}

void bcast_strides(const std::vector<int64_t>& in, const std::vector<int64_t>& os, int64_t* dst) {
    const std::vector<int64_t> st = strides(in);
    const size_t r = os.size();
    for (int d = 0; d < RANK; ++d) dst[d] = 0;
    for (size_t i = 0; i < in.size(); ++i) dst[RANK - in.size() + i] = in[i] == 1 ? 0 : st[i];
    (void)r;
}

}  // namespace

bool TkOnnxExec::exec_seq_op(const TkOnnxNode& nd, std::map<std::string, Val>& v, bool* handled) {
    auto in = [&](size_t i) -> Val* {
        if (i >= nd.in.size() || nd.in[i].empty()) return nullptr;
        auto it = v.find(nd.in[i]);
        return it == v.end() ? nullptr : &it->second;
    };
    auto need = [&](size_t i) -> Val* {
        Val* x = in(i);
        if (!x) { error = "input " + std::to_string(i) + " is missing"; return nullptr; }
        if (x->is_int) { error = "input " + std::to_string(i) + " must be a float tensor"; return nullptr; }
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
    auto out_i = [&](const std::vector<int64_t>& shape, std::vector<int64_t>&& data) {
        Val o;
        o.is_int = true;
        o.shape = shape;
        o.ints = std::move(data);
        v[nd.out[0]] = o;
    };
    const std::string& op = nd.op;
    *handled = true;

    if (op == "Shape") {
        Val* x = in(0);
        if (!x) { error = "input 0 is missing"; return false; }
        const int64_t r = (int64_t)x->shape.size();
        int64_t s = nd.ai("start", 0), e = nd.has("end") ? nd.ai("end", r) : r;
        if (s < 0) s += r;
        if (e < 0) e += r;
        s = std::min(std::max<int64_t>(s, 0), r);
        e = std::min(std::max<int64_t>(e, s), r);
        out_i({e - s}, std::vector<int64_t>(x->shape.begin() + s, x->shape.begin() + e));
        return true;
    }
    if (op == "If") {
        /* the condition is host data here (a comparison of the integer sample-rate input in VAD graphs of the Silero class; a bool
         * initialiser): the chosen branch's nodes run in this value map — a sub-graph sees the outer scope's names, ONNX IR "Graphs",
         * name scoping — and its declared outputs become the node's */
        Val* c = in(0);
        if (!c) { error = "condition is missing"; return false; }
        bool take;
        if (c->is_int && c->ints.size() == 1) take = c->ints[0] != 0;
        else if (!c->is_int && c->host.size() == 1) take = c->host[0] != 0.0f;
        else { error = "the condition must be a scalar known on the host (an integer / bool tensor or a small float constant)"; return false; }
        auto br = nd.attr.find(take ? "then_branch" : "else_branch");
        if (br == nd.attr.end() || !br->second.g) { error = "branch graph missing"; return false; }
        const TkOnnxGraph& sub = *br->second.g;
        if (sub.outputs.size() != nd.out.size()) { error = "the branch declares another number of outputs than the node"; return false; }
        for (const auto& sn : sub.nodes) {
            if (sn.op == "Constant") continue;
            if (!exec(sn, v)) { if (error.find(sn.op) == std::string::npos) error = sn.op + " (node '" + sn.name + "' of a branch): " + error; return false; }
        }
        for (size_t i = 0; i < nd.out.size(); ++i) {
            auto it = v.find(sub.outputs[i].name);
            if (it == v.end()) { error = "branch output '" + sub.outputs[i].name + "' was not produced"; return false; }
            const Val o = it->second;
            v[nd.out[i]] = o;
        }
        return true;
    }
    if (op == "Loop" || op == "Scan") {
        /* Loop (a counted / conditional repetition of a body graph with loop-carried values and per-iteration "scan" outputs) and Scan (a body
         * graph mapped over the slices of its scan inputs along axis 0, with state values): what exporters emit for recurrences written as
         * Python loops.  Trip count and condition are host data, like If's condition (an integer / bool tensor, possibly computed by the body
         * from the iteration number).  The body's nodes run in this value map (ONNX name scoping: a sub-graph sees the outer scope), every
         * iteration's outputs are fresh arena tensors, carried values are rebound by name.  Scan outputs are stacked along a new (Loop) or the
         * first (Scan) axis.  Not covered: scan axes other than 0, reversed scan OUTPUTS, sequence-typed carried values. */
        auto bd = nd.attr.find("body");
        if (bd == nd.attr.end() || !bd->second.g) { error = "body graph missing"; return false; }
        const TkOnnxGraph& body = *bd->second.g;
        const bool is_loop = op == "Loop";
        auto host_flag = [&](const Val& c, bool* flag) {
            if (c.is_int && c.ints.size() == 1) { *flag = c.ints[0] != 0; return true; }
            if (!c.is_int && c.host.size() == 1) { *flag = c.host[0] != 0.0f; return true; }
            return false;
        };
        auto run_body = [&]() {
            for (const auto& sn : body.nodes) {
                if (sn.op == "Constant") continue;
                if (!exec(sn, v)) { if (error.find(sn.op) == std::string::npos) error = sn.op + " (node '" + sn.name + "' of a loop body): " + error; return false; }
            }
            return true;
        };
        auto fetch = [&](const std::string& name, Val* dst) {
            auto it = v.find(name);
            if (it == v.end()) { error = "body output '" + name + "' was not produced"; return false; }
            *dst = it->second;
            return true;
        };
        /* stack the per-iteration values along a new leading axis (Loop and Scan alike) */
        auto stack = [&](const std::vector<Val>& parts, const std::string& name) {
            if (name.empty()) return true;
            Val o;
            if (parts.empty()) { o.shape = {0}; o.is_int = true; v[name] = o; return true; }
            o.shape = parts[0].shape;
            o.shape.insert(o.shape.begin(), (int64_t)parts.size());
            o.is_int = parts[0].is_int;
            const int64_t each = parts[0].count();
            for (const Val& p : parts)
                if (p.shape != parts[0].shape || p.is_int != o.is_int) { error = "a scan output changes its shape between iterations"; return false; }
            if (o.is_int) {
                for (const Val& p : parts) o.ints.insert(o.ints.end(), p.ints.begin(), p.ints.end());
            } else {
                o.d = alloc(each * (int64_t)parts.size());
                if (!o.d) { error = "activation arena exhausted"; return false; }
                for (size_t i = 0; i < parts.size(); ++i)
                    if (each > 0) SQ(hipMemcpyAsync(o.d + (int64_t)i * each, parts[i].d, (size_t)each * 4, hipMemcpyDeviceToDevice, stream_));
            }
            v[name] = o;
            return true;
        };
        const int64_t kMaxIter = 1 << 16; /* a run's activations live in one arena: a loop that long has exhausted it long before */
        if (is_loop) {
            if (nd.in.size() < 2) { error = "Loop needs the trip-count and condition inputs (either may be empty)"; return false; }
            const size_t N = nd.in.size() - 2;
            if (body.inputs.size() != N + 2 || body.outputs.size() < N + 1) { error = "the body's inputs / outputs do not match the node's loop-carried values"; return false; }
            const size_t K = body.outputs.size() - 1 - N;
            if (nd.out.size() != N + K) { error = "the node declares another number of outputs than carried values + scan outputs"; return false; }
            int64_t trip = kMaxIter + 1;
            bool keep = true;
            if (Val* m = in(0)) {
                if (!m->is_int || m->ints.size() != 1) { error = "the trip count must be an integer scalar known on the host"; return false; }
                trip = m->ints[0];
            }
            if (Val* c = in(1)) {
                if (!host_flag(*c, &keep)) { error = "the condition must be a scalar known on the host"; return false; }
            }
            std::vector<Val> carried(N);
            for (size_t i = 0; i < N; ++i) {
                Val* x = in(2 + i);
                if (!x) { error = "loop-carried input " + std::to_string(i) + " is missing"; return false; }
                carried[i] = *x;
            }
            std::vector<std::vector<Val>> scans(K);
            for (int64_t it = 0; it < trip && keep; ++it) {
                if (it >= kMaxIter) { error = "more than 65536 iterations"; return false; }
                Val iv; iv.is_int = true; iv.ints = {it};
                Val cv; cv.is_int = true; cv.ints = {1};
                v[body.inputs[0].name] = iv;
                v[body.inputs[1].name] = cv;
                for (size_t i = 0; i < N; ++i) v[body.inputs[2 + i].name] = carried[i];
                if (!run_body()) return false;
                Val co;
                if (!fetch(body.outputs[0].name, &co)) return false;
                if (!host_flag(co, &keep)) { error = "the body's condition output must be a scalar known on the host (integer / bool arithmetic on the iteration number)"; return false; }
                std::vector<Val> next(N);
                for (size_t i = 0; i < N; ++i) if (!fetch(body.outputs[1 + i].name, &next[i])) return false;
                carried.swap(next);
                for (size_t k = 0; k < K; ++k) { Val s; if (!fetch(body.outputs[1 + N + k].name, &s)) return false; scans[k].push_back(s); }
            }
            for (size_t i = 0; i < N; ++i) if (!nd.out[i].empty()) v[nd.out[i]] = carried[i];
            for (size_t k = 0; k < K; ++k) if (!stack(scans[k], nd.out[N + k])) return false;
            return true;
        }
        /* Scan (opset >= 9: no sequence_lens input) */
        const int64_t M = nd.ai("num_scan_inputs", 0);
        if (M < 1 || (size_t)M > nd.in.size()) { error = "num_scan_inputs is missing or larger than the input count"; return false; }
        const size_t N = nd.in.size() - (size_t)M;
        if (body.inputs.size() != nd.in.size() || body.outputs.size() < N) { error = "the body's inputs / outputs do not match the node's state and scan inputs"; return false; }
        const size_t K = body.outputs.size() - N;
        if (nd.out.size() != N + K) { error = "the node declares another number of outputs than state values + scan outputs"; return false; }
        for (const char* a : {"scan_input_axes", "scan_output_axes"})
            if (const std::vector<int64_t>* ax = nd.aints(a))
                for (int64_t x : *ax) if (x != 0) { error = std::string(a) + " other than 0"; return false; }
        if (const std::vector<int64_t>* od = nd.aints("scan_output_directions"))
            for (int64_t x : *od) if (x != 0) { error = "reversed scan outputs"; return false; }
        const std::vector<int64_t>* idir = nd.aints("scan_input_directions");
        std::vector<Val> state(N), xs((size_t)M);
        for (size_t i = 0; i < N; ++i) {
            Val* x = in(i);
            if (!x) { error = "state input " + std::to_string(i) + " is missing"; return false; }
            state[i] = *x;
        }
        int64_t T = -1;
        for (size_t j = 0; j < (size_t)M; ++j) {
            Val* x = in(N + j);
            if (!x || x->shape.empty()) { error = "scan input " + std::to_string(j) + " is missing or a scalar"; return false; }
            if (T >= 0 && x->shape[0] != T) { error = "the scan inputs disagree in their length"; return false; }
            T = x->shape[0];
            xs[j] = *x;
        }
        if (T > kMaxIter) { error = "more than 65536 iterations"; return false; }
        std::vector<std::vector<Val>> scans(K);
        for (int64_t t = 0; t < T; ++t) {
            for (size_t i = 0; i < N; ++i) v[body.inputs[i].name] = state[i];
            for (size_t j = 0; j < (size_t)M; ++j) { /* slice t (or T - 1 - t) of scan input j: a view, the tensors are row-major */
                const bool rev = idir && j < idir->size() && (*idir)[j] == 1;
                const int64_t at = rev ? T - 1 - t : t;
                Val sl;
                sl.shape.assign(xs[j].shape.begin() + 1, xs[j].shape.end());
                const int64_t each = sl.count();
                sl.is_int = xs[j].is_int;
                if (sl.is_int) sl.ints.assign(xs[j].ints.begin() + at * each, xs[j].ints.begin() + (at + 1) * each);
                else sl.d = xs[j].d + at * each;
                v[body.inputs[N + j].name] = sl;
            }
            if (!run_body()) return false;
            std::vector<Val> next(N);
            for (size_t i = 0; i < N; ++i) if (!fetch(body.outputs[i].name, &next[i])) return false;
            state.swap(next);
            for (size_t k = 0; k < K; ++k) { Val s; if (!fetch(body.outputs[N + k].name, &s)) return false; scans[k].push_back(s); }
        }
        for (size_t i = 0; i < N; ++i) if (!nd.out[i].empty()) v[nd.out[i]] = state[i];
        for (size_t k = 0; k < K; ++k) if (!stack(scans[k], nd.out[N + k])) return false;
        return true;
    }
    if ((op == "Equal" || op == "Less" || op == "Greater") && in(0) && in(1) && (!in(0)->is_int || !in(1)->is_int)) {
        /* a comparison of small FLOAT tensors — the data-dependent condition of an If / Loop ("is the largest activation above a threshold?"): both
         * sides are brought to the host (a stream synchronisation: the price of control flow decided by device data) and the result is host
         * data like every other condition.  Up to 64 elements; larger float comparisons are masks, not conditions, and are not covered. */
        auto fetch_small = [&](Val* x, std::vector<float>* dst) {
            const int64_t n = x->is_int ? (int64_t)x->ints.size() : x->count();
            if (n < 1 || n > 64) { error = "float comparisons are evaluated on the host for up to 64 elements (conditions, not masks)"; return false; }
            dst->resize((size_t)n);
            if (x->is_int) { for (int64_t i = 0; i < n; ++i) (*dst)[(size_t)i] = (float)x->ints[(size_t)i]; return true; }
            if ((int64_t)x->host.size() == n) { *dst = x->host; return true; }
            SQ(hipMemcpyAsync(dst->data(), x->d, (size_t)n * 4, hipMemcpyDeviceToHost, stream_));
            SQ(hipStreamSynchronize(stream_));
            return true;
        };
        std::vector<float> fa, fb;
        Val ac = *in(0), bc = *in(1);
        if (!fetch_small(&ac, &fa) || !fetch_small(&bc, &fb)) return false;
        const size_t na = fa.size(), nb = fb.size();
        if (na != nb && na != 1 && nb != 1) { error = "operands do not broadcast"; return false; }
        std::vector<int64_t> o(std::max(na, nb));
        for (size_t i = 0; i < o.size(); ++i) {
            const float x = fa[na == 1 ? 0 : i], y = fb[nb == 1 ? 0 : i];
            o[i] = op == "Equal" ? x == y : op == "Less" ? x < y : x > y;
        }
        out_i(na >= nb ? ac.shape : bc.shape, std::move(o));
        return true;
    }
    if (op == "Equal" || op == "Less" || op == "Greater" || op == "And" || op == "Or" || op == "Not") {
        Val* a = in(0);
        if (!a || !a->is_int) { error = "logic runs on host integer / bool tensors only"; return false; }
        const Val ac = *a;
        std::vector<int64_t> o(ac.ints.size());
        if (op == "Not") {
            for (size_t i = 0; i < o.size(); ++i) o[i] = ac.ints[i] ? 0 : 1;
            out_i(ac.shape, std::move(o));
            return true;
        }
        Val* b = in(1);
        if (!b || !b->is_int) { error = "comparisons and logic run on host integer / bool tensors only"; return false; }
        const Val bc = *b;
        const size_t na = ac.ints.size(), nb = bc.ints.size();
        if (na != nb && na != 1 && nb != 1) { error = "integer operands do not broadcast"; return false; }
        o.resize(std::max(na, nb));
        for (size_t i = 0; i < o.size(); ++i) {
            const int64_t x = ac.ints[na == 1 ? 0 : i], y = bc.ints[nb == 1 ? 0 : i];
            o[i] = op == "Equal" ? x == y : op == "Less" ? x < y : op == "Greater" ? x > y : op == "And" ? (x && y) : (x || y);
        }
        out_i(na >= nb ? ac.shape : bc.shape, std::move(o));
        return true;
    }
    /* integer tensors (shape sub-graphs): evaluated on the host */
    if ((op == "Add" || op == "Sub" || op == "Mul" || op == "Div") && in(0) && in(1) && in(0)->is_int && in(1)->is_int) {
        const Val a = *in(0), b = *in(1);
        const size_t na = a.ints.size(), nb = b.ints.size();
        if (na != nb && na != 1 && nb != 1) { error = "integer operands do not broadcast"; return false; }
        const size_t n = std::max(na, nb);
        std::vector<int64_t> o(n);
        for (size_t i = 0; i < n; ++i) {
            const int64_t x = a.ints[na == 1 ? 0 : i], y = b.ints[nb == 1 ? 0 : i];
            if (op == "Div" && y == 0) { error = "integer division by zero"; return false; }
            o[i] = op == "Add" ? x + y : op == "Sub" ? x - y : op == "Mul" ? x * y : x / y;
        }
        out_i(na >= nb ? a.shape : b.shape, std::move(o));
        return true;
    }
    if (op == "Concat" && in(0) && in(0)->is_int) {
        std::vector<int64_t> o;
        for (size_t i = 0; i < nd.in.size(); ++i) {
            Val* p = in(i);
            if (!p || !p->is_int || p->shape.size() > 1) { error = "integer Concat takes 1-D integer tensors"; return false; }
            o.insert(o.end(), p->ints.begin(), p->ints.end());
        }
        const int64_t n = (int64_t)o.size();
        out_i({n}, std::move(o));
        return true;
    }
    if (op == "Gather") {
        Val* x = in(0);
        Val* ix = in(1);
        if (!x || !ix) { error = "inputs are missing"; return false; }
        if (!ix->is_int) { error = "indices must be an integer tensor"; return false; }
        const Val xc = *x, ic = *ix;
        const int64_t r = (int64_t)xc.shape.size();
        int64_t ax = nd.ai("axis", 0);
        if (ax < 0) ax += r;
        if (ax < 0 || ax >= r) { error = "Gather axis outside the rank of its data"; return false; }
        const int64_t dim = xc.shape[(size_t)ax];
        std::vector<int32_t> idx(ic.ints.size());
        for (size_t i = 0; i < idx.size(); ++i) {
            int64_t t = ic.ints[i] < 0 ? ic.ints[i] + dim : ic.ints[i];
            if (t < 0 || t >= dim) { error = "Gather index out of range"; return false; }
            idx[i] = (int32_t)t;
        }
        std::vector<int64_t> os(xc.shape.begin(), xc.shape.begin() + ax);
        os.insert(os.end(), ic.shape.begin(), ic.shape.end());
        os.insert(os.end(), xc.shape.begin() + ax + 1, xc.shape.end());
        if (xc.is_int) {
            if (r != 1) { error = "integer Gather takes 1-D data"; return false; }
            std::vector<int64_t> o(idx.size());
            for (size_t i = 0; i < idx.size(); ++i) o[i] = xc.ints[(size_t)idx[i]];
            out_i(ic.shape, std::move(o));
            return true;
        }
        int64_t outer = 1, inner = 1;
        for (int64_t i = 0; i < ax; ++i) outer *= xc.shape[(size_t)i];
        for (int64_t i = ax + 1; i < r; ++i) inner *= xc.shape[(size_t)i];
        Val* y = out_f(0, os);
        if (!y) return false;
        if (y->count() == 0) return true;
        float* didx = alloc((int64_t)idx.size());
        if (!didx) { error = "activation arena exhausted"; return false; }
        staging_.push_back(std::move(idx)); /* stays alive until the next begin(): the copy below may still read it */
        SQ(hipMemcpyAsync(didx, staging_.back().data(), staging_.back().size() * 4, hipMemcpyHostToDevice, stream_));
        hipLaunchKernelGGL(k_sq_gather, grid_of(y->count()), dim3(128), 0, stream_, xc.d, (const int32_t*)didx, y->d, outer, dim, (int64_t)staging_.back().size(), inner);
        return true;
    }
    if (op == "Erf" || op == "Gelu") {
        Val* x = need(0);
        if (!x) return false;
        const Val xc = *x;
        int mode = E_ERF;
        if (op == "Gelu") {
            const std::string ap = nd.as("approximate", "none");
            if (ap != "none" && ap != "tanh") { error = "Gelu approximate='" + ap + "' is not supported"; return false; }
            mode = ap == "tanh" ? E_GELU_TANH : E_GELU;
        }
        Val* y = out_f(0, xc.shape);
        if (!y) return false;
        if (xc.count() > 0) hipLaunchKernelGGL(k_sq_erf, grid_of(xc.count()), dim3(128), 0, stream_, mode, xc.d, y->d, xc.count());
        return true;
    }
    if (op == "LayerNormalization") {
        Val* x = need(0);
        Val* sc = x ? need(1) : nullptr;
        if (!sc) return false;
        const Val xc = *x, scc = *sc;
        const int64_t r = (int64_t)xc.shape.size();
        int64_t ax = nd.ai("axis", -1);
        if (ax < 0) ax += r;
        if (r < 1 || ax != r - 1) { error = "only normalisation over the last axis"; return false; }
        const int D = (int)xc.shape.back();
        if (scc.count() != D) { error = "Scale must hold one value per element of the last axis"; return false; }
        const float* bias = nullptr;
        if (Val* b = in(2)) { if (b->is_int || b->count() != D) { error = "B must hold one value per element of the last axis"; return false; } bias = b->d; }
        if (nd.out.size() > 1 && !nd.out[1].empty()) { error = "the Mean / InvStdDev outputs are not produced"; return false; }
        if (!bias) {
            float* z = alloc(D);
            if (!z) { error = "activation arena exhausted"; return false; }
            hipLaunchKernelGGL(k_sq_fill, grid_of(D), dim3(128), 0, stream_, z, 0.0f, (int64_t)D);
            bias = z;
        }
        Val* y = out_f(0, xc.shape);
        if (!y) return false;
        if (xc.count() > 0) tk_launch_layernorm(xc.d, (int)(xc.count() / D), D, scc.d, bias, nd.af("epsilon", 1e-5f), y->d, stream_);
        return true;
    }
    if (op == "ReduceSum" || op == "ReduceL2" || op == "ReduceMax" || op == "ReduceMin") {
        Val* x = need(0);
        if (!x) return false;
        const Val xc = *x;
        std::vector<int64_t> axes;
        if (const std::vector<int64_t>* a = nd.aints("axes")) axes = *a;
        else if (Val* a1 = in(1)) { if (a1->is_int) axes = a1->ints; }
        if (axes.empty()) { if (nd.ai("noop_with_empty_axes", 0)) { v[nd.out[0]] = xc; return true; } for (size_t i = 0; i < xc.shape.size(); ++i) axes.push_back((int64_t)i); }
        for (auto& a : axes) if (a < 0) a += (int64_t)xc.shape.size();
        std::sort(axes.begin(), axes.end());
        for (size_t i = 1; i < axes.size(); ++i) if (axes[i] != axes[i - 1] + 1) { error = "reduced axes must be adjacent"; return false; }
        if (axes.front() < 0 || axes.back() >= (int64_t)xc.shape.size()) { error = "reduction axis outside the rank of its input"; return false; }
        int64_t outer = 1, mid = 1, inner = 1;
        std::vector<int64_t> os;
        const bool keep = nd.ai("keepdims", 1) != 0;
        for (size_t i = 0; i < xc.shape.size(); ++i) {
            if ((int64_t)i < axes.front()) { outer *= xc.shape[i]; os.push_back(xc.shape[i]); }
            else if ((int64_t)i <= axes.back()) { mid *= xc.shape[i]; if (keep) os.push_back(1); }
            else { inner *= xc.shape[i]; os.push_back(xc.shape[i]); }
        }
        if (mid < 1) { error = "reduction over an empty axis"; return false; }
        Val* y = out_f(0, os);
        if (!y) return false;
        const int rop = op == "ReduceSum" ? R_SUM : op == "ReduceL2" ? R_L2 : op == "ReduceMax" ? R_MAX : R_MIN;
        if (outer * inner > 0) hipLaunchKernelGGL(k_sq_reduce, grid_of(outer * inner), dim3(128), 0, stream_, rop, xc.d, y->d, outer, mid, inner);
        return true;
    }
    if (op == "Max" || op == "Min" || op == "Expand" || op == "Where") {
        const size_t ia = op == "Where" ? 1 : 0, ib = op == "Where" ? 2 : 1;
        Val* a = need(ia);
        if (!a) return false;
        const Val ac = *a;
        Val bc = ac, cc = ac;
        std::vector<std::vector<int64_t>> shapes{ac.shape};
        if (op == "Expand") {
            Val* s = in(1);
            if (!s || !s->is_int) { error = "shape must be an integer tensor"; return false; }
            for (int64_t d : s->ints) if (d < 1) { error = "Expand to a dimension below 1"; return false; }
            shapes.push_back(s->ints);
        } else {
            if (nd.in.size() != (op == "Where" ? 3u : 2u)) { error = "takes exactly " + std::string(op == "Where" ? "three" : "two") + " inputs here"; return false; }
            Val* b = need(ib);
            if (!b) return false;
            bc = *b;
            shapes.push_back(bc.shape);
            if (op == "Where") {
                Val* c = in(0);
                if (!c) { error = "condition is missing"; return false; }
                cc = *c;
                if (cc.is_int) { /* a constant boolean / integer mask: as floats on the device */
                    float* d = alloc((int64_t)cc.ints.size());
                    if (!d) { error = "activation arena exhausted"; return false; }
                    std::vector<int32_t> bits(cc.ints.size());
                    for (size_t i = 0; i < bits.size(); ++i) { const float f = cc.ints[i] ? 1.0f : 0.0f; memcpy(&bits[i], &f, 4); }
                    staging_.push_back(std::move(bits));
                    SQ(hipMemcpyAsync(d, staging_.back().data(), staging_.back().size() * 4, hipMemcpyHostToDevice, stream_));
                    cc.d = d;
                }
                shapes.push_back(cc.shape);
            }
        }
        std::vector<int64_t> os;
        if (!broadcast_shape(shapes, &os)) { error = "shapes do not broadcast"; return false; }
        if (os.size() > (size_t)RANK) { error = "rank > 6"; return false; }
        SqIdx ix{};
        for (int d = 0; d < RANK; ++d) ix.dim[d] = 1;
        for (size_t i = 0; i < os.size(); ++i) ix.dim[RANK - os.size() + i] = os[i];
        bcast_strides(ac.shape, os, ix.sa);
        bcast_strides(bc.shape, os, ix.sb);
        bcast_strides(cc.shape, os, ix.sc);
        Val* y = out_f(0, os);
        if (!y) return false;
        const int bop = op == "Max" ? 0 : op == "Min" ? 1 : op == "Expand" ? 2 : 3;
        if (y->count() > 0) hipLaunchKernelGGL(k_sq_bcast, grid_of(y->count()), dim3(128), 0, stream_, bop, ac.d, bc.d, cc.d, y->d, ix, y->count());
        return true;
    }
    if (op == "MatMul" && in(0) && in(1) && !in(0)->is_int && !in(1)->is_int && in(1)->shape.size() > 2) {
        /* two activations (QK^T, PV): [..., M, K] x [..., K, N] with equal leading dimensions, or B's all 1 */
        const Val ac = *in(0), bc = *in(1);
        const size_t ra = ac.shape.size(), rb = bc.shape.size();
        if (ra < 2 || ra != rb) { error = "batched MatMul expects operands of equal rank"; return false; }
        const int M = (int)ac.shape[ra - 2], K = (int)ac.shape[ra - 1], N = (int)bc.shape[rb - 1];
        if (bc.shape[rb - 2] != K) { error = "inner dimensions differ"; return false; }
        int64_t batch = 1, bbatch = 1;
        for (size_t i = 0; i + 2 < ra; ++i) { batch *= ac.shape[i]; bbatch *= bc.shape[i]; if (bc.shape[i] != ac.shape[i] && bc.shape[i] != 1) { error = "leading dimensions differ"; return false; } }
        if (bbatch != batch && bbatch != 1) { error = "partially broadcast leading dimensions are not supported"; return false; }
        std::vector<int64_t> os = ac.shape;
        os.back() = N;
        Val* y = out_f(0, os);
        if (!y) return false;
        if (y->count() == 0) return true;
        TkGemm g{};
        g.A = ac.d; g.B = bc.d; g.C = y->d;
        g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = N; g.ldc = N;
        g.b_kn = 1; g.act = TK_ACT_NONE; g.alpha = 1.0f;
        g.batch = (int)batch; g.sA = (int64_t)M * K; g.sB = bbatch == 1 ? 0 : (int64_t)K * N; g.sC = (int64_t)M * N;
        tk_launch_gemm(g, stream_);
        return true;
    }
    if (op == "ConvTranspose") {
        Val* x = need(0);
        Val* w = x ? need(1) : nullptr;
        if (!w) return false;
        const Val xc = *x, wc = *w;
        if (xc.shape.size() != 4 || wc.shape.size() != 4) { error = "expects X [N, C, H, W] and W [C, M, kh, kw]"; return false; }
        if (nd.ai("group", 1) != 1) { error = "grouped ConvTranspose is not supported"; return false; }
        if (const auto* dl = nd.aints("dilations")) for (int64_t d : *dl) if (d != 1) { error = "dilated ConvTranspose is not supported"; return false; }
        if (nd.as("auto_pad", "NOTSET") != "NOTSET") { error = "auto_pad is not supported"; return false; }
        CtParams p{};
        p.N = (int)xc.shape[0]; p.Ci = (int)xc.shape[1]; p.H = (int)xc.shape[2]; p.W = (int)xc.shape[3];
        p.Co = (int)wc.shape[1]; p.kh = (int)wc.shape[2]; p.kw = (int)wc.shape[3];
        if (wc.shape[0] != p.Ci) { error = "weight channels differ from the input's"; return false; }
        p.sh = p.sw = 1;
        if (const auto* s = nd.aints("strides")) if (s->size() == 2) { p.sh = (int)(*s)[0]; p.sw = (int)(*s)[1]; }
        int pb = 0, pr = 0, oph = 0, opw = 0;
        if (const auto* pd = nd.aints("pads")) if (pd->size() == 4) { p.pt = (int)(*pd)[0]; p.pl = (int)(*pd)[1]; pb = (int)(*pd)[2]; pr = (int)(*pd)[3]; }
        if (const auto* op2 = nd.aints("output_padding")) if (op2->size() == 2) { oph = (int)(*op2)[0]; opw = (int)(*op2)[1]; }
        if (p.sh < 1 || p.sw < 1 || p.kh < 1 || p.kw < 1 || p.pt < 0 || p.pl < 0 || pb < 0 || pr < 0) { error = "bad strides / pads"; return false; }
        p.Ho = (p.H - 1) * p.sh + p.kh - p.pt - pb + oph;
        p.Wo = (p.W - 1) * p.sw + p.kw - p.pl - pr + opw;
        if (const auto* osz = nd.aints("output_shape")) if (osz->size() == 2 && ((int)(*osz)[0] != p.Ho || (int)(*osz)[1] != p.Wo)) { error = "output_shape differs from what strides and pads give"; return false; }
        if (p.Ho < 1 || p.Wo < 1) { error = "empty output"; return false; }
        const float* bias = nullptr;
        if (Val* b = in(2)) { if (b->is_int || b->count() != p.Co) { error = "B must hold one value per output channel"; return false; } bias = b->d; }
        Val* y = out_f(0, {p.N, p.Co, p.Ho, p.Wo});
        if (!y) return false;
        hipLaunchKernelGGL(k_sq_conv_transpose, grid_of(y->count()), dim3(128), 0, stream_, xc.d, wc.d, bias, y->d, p, y->count());
        return true;
    }
    if (op == "Slice" && in(0) && !in(0)->is_int) {
        /* only the strided form is taken here (steps > 1); unit steps stay with tk_onnx_exec.hip */
        Val* st = in(4);
        bool strided = false;
        if (st && st->is_int) for (int64_t s : st->ints) strided = strided || s != 1;
        if (!strided) { *handled = false; return true; }
        const Val xc = *in(0);
        const size_t r = xc.shape.size();
        if (r > (size_t)RANK) { error = "rank > 6"; return false; }
        Val *s0 = in(1), *e0 = in(2), *a0 = in(3);
        if (!s0 || !e0 || !s0->is_int || !e0->is_int) { error = "starts / ends are missing"; return false; }
        std::vector<int64_t> axes;
        if (a0 && a0->is_int) axes = a0->ints;
        else for (size_t i = 0; i < s0->ints.size(); ++i) axes.push_back((int64_t)i);
        if (axes.size() != s0->ints.size() || axes.size() != e0->ints.size() || axes.size() != st->ints.size()) { error = "starts / ends / axes / steps differ in length"; return false; }
        const std::vector<int64_t> sx = strides(xc.shape);
        std::vector<int64_t> os = xc.shape, step(r, 1), begin(r, 0);
        for (size_t i = 0; i < axes.size(); ++i) {
            const int64_t a = axes[i] < 0 ? axes[i] + (int64_t)r : axes[i];
            if (a < 0 || a >= (int64_t)r) { error = "Slice axis outside the rank of its input"; return false; }
            if (st->ints[i] < 1) { error = "only positive steps are supported"; return false; }
            const int64_t dim = xc.shape[(size_t)a];
            int64_t s = s0->ints[i] < 0 ? s0->ints[i] + dim : s0->ints[i], e = e0->ints[i] < 0 ? e0->ints[i] + dim : e0->ints[i];
            s = std::min(std::max<int64_t>(s, 0), dim);
            e = std::min(std::max<int64_t>(e, 0), dim);
            begin[(size_t)a] = s;
            step[(size_t)a] = st->ints[i];
            os[(size_t)a] = e > s ? (e - s + st->ints[i] - 1) / st->ints[i] : 0;
        }
        SqIdx ix{};
        for (int d = 0; d < RANK; ++d) { ix.dim[d] = 1; ix.sa[d] = 0; }
        int64_t off = 0;
        for (size_t i = 0; i < r; ++i) { const size_t d = RANK - r + i; ix.dim[d] = os[i]; ix.sa[d] = sx[i] * step[i]; off += begin[i] * sx[i]; }
        Val* y = out_f(0, os);
        if (!y) return false;
        if (y->count() > 0) hipLaunchKernelGGL(k_sq_strided, grid_of(y->count()), dim3(128), 0, stream_, xc.d, y->d, ix, off, y->count());
        return true;
    }
    *handled = false;
    return true;
}
