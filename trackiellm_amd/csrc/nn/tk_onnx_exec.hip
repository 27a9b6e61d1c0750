/* tk_onnx_exec.hip — see tk_onnx_exec.h */
#include "tk_onnx_exec.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>

#include "../common/tk_exact_math.h"
#include "tk_nn_kernels.h"

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

#define TK_VG_RANK 6 /* window partition / merge of a Swin block is a rank-6 transpose */
struct VgIdx { int64_t dim[TK_VG_RANK], sa[TK_VG_RANK], sb[TK_VG_RANK]; };

__global__ void k_vg_binary(int op, const float* a, const float* b, float* y, VgIdx ix, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t r = i, oa = 0, ob = 0;
    for (int d = TK_VG_RANK - 1; d >= 0; --d) {
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

/* softmax over an inner axis of n entries `inner` elements apart: max, exp and sum in index order, divide */
__global__ void k_oe_softmax_axis(const float* x, float* y, int64_t pairs, int64_t n, int64_t inner) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pairs) return;
    const int64_t base = (i / inner) * n * inner + i % inner;
    float m = -INFINITY;
    for (int64_t k = 0; k < n; ++k) m = tk_fmaxf(m, x[base + k * inner]);
    float s = 0.0f;
    for (int64_t k = 0; k < n; ++k) { const float e = tk_expf(x[base + k * inner] - m); y[base + k * inner] = e; s = s + e; }
    for (int64_t k = 0; k < n; ++k) y[base + k * inner] = tk_divf(y[base + k * inner], s);
}

/* generic gather: out element i (coordinates over dim[]) reads x[off + sum c_d * sa[d]] — Transpose, Slice, Concat pieces */
__global__ void k_vg_gather(const float* x, float* y, VgIdx ix, int64_t off, int64_t n, int64_t y_off, VgIdx oy) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t r = i, oa = off, oo = y_off;
    for (int d = TK_VG_RANK - 1; d >= 0; --d) {
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


/* ------------------------------------------------------------------ kernels of the 2-D image graphs ---- */

enum { P_LEAKY, P_HSIGMOID, P_HSWISH, P_CLIP };

__global__ void k_oe_unary_p(int op, const float* x, float* y, int64_t n, float p0, float p1) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    float r;
    switch (op) {
        case P_LEAKY: r = v >= 0.0f ? v : v * p0; break;
        case P_HSIGMOID: r = tk_fmaxf(0.0f, tk_fminf(1.0f, tk_fmaf(p0, v, p1))); break;
        case P_HSWISH: r = v * tk_fmaxf(0.0f, tk_fminf(1.0f, tk_fmaf(1.0f / 6.0f, v, 0.5f))); break;
        default: r = tk_fminf(tk_fmaxf(v, p0), p1); break; /* Clip: max(min) first, as ONNX's reference does */
    }
    y[i] = r;
}

struct OeConv { int C, H, W, M, kh, kw, sh, sw, pt, pl, dh, dw, Ho, Wo, groups; };

/* col[o][k]: o = ho * Wo + wo, k = (c * kh + i) * kw + j; column K = 1 (the bias rides as one more product), columns K + 1 .. K1 - 1 = 0 */
__global__ void k_oe_im2col(const float* x, OeConv p, float* col, int K, int K1) {
    const int64_t total = (int64_t)p.Ho * p.Wo * K1;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(e % K1);
        const int64_t o = e / K1;
        float v = 0.0f;
        if (k < K) {
            const int j = k % p.kw, i = (k / p.kw) % p.kh, c = k / (p.kw * p.kh);
            const int ho = (int)(o / p.Wo), wo = (int)(o % p.Wo);
            const int h = ho * p.sh - p.pt + i * p.dh, w = wo * p.sw - p.pl + j * p.dw;
            if (h >= 0 && h < p.H && w >= 0 && w < p.W) v = x[((int64_t)c * p.H + h) * p.W + w];
        } else if (k == K) v = 1.0f;
        col[e] = v;
    }
}

/* grouped / depthwise convolution, one thread per output element: the same chain (channel of the group outer, kernel row, kernel column
 * inner, bias last) as the dense path */
__global__ void k_oe_conv2d_direct(const float* x, const float* w, const float* b, float* y, OeConv p, int N) {
    const int64_t total = (int64_t)N * p.M * p.Ho * p.Wo;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int wo = (int)(i % p.Wo), ho = (int)((i / p.Wo) % p.Ho), m = (int)((i / ((int64_t)p.Wo * p.Ho)) % p.M), n = (int)(i / ((int64_t)p.Wo * p.Ho * p.M));
    const int cg = p.C / p.groups, mg = p.M / p.groups, g = m / mg;
    float acc = 0.0f;
    for (int c = 0; c < cg; ++c) {
        const float* xc = x + ((int64_t)n * p.C + g * cg + c) * p.H * p.W;
        const float* wc = w + ((int64_t)m * cg + c) * p.kh * p.kw;
        for (int a = 0; a < p.kh; ++a) {
            const int h = ho * p.sh - p.pt + a * p.dh;
            if (h < 0 || h >= p.H) continue;
            for (int q = 0; q < p.kw; ++q) {
                const int ww = wo * p.sw - p.pl + q * p.dw;
                if (ww >= 0 && ww < p.W) acc = tk_fmaf(xc[(int64_t)h * p.W + ww], wc[a * p.kw + q], acc);
            }
        }
    }
    y[i] = b ? acc + b[m] : acc;
}

__global__ void k_oe_pack_conv(const float* w, const float* b, float* wp, int M, int K, int K1) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * K1) return;
    const int k = (int)(i % K1), m = (int)(i / K1);
    wp[i] = k < K ? w[(int64_t)m * K + k] : (k == K && b) ? b[m] : 0.0f;
}

enum { CT_HALF_PIXEL, CT_PYTORCH_HALF_PIXEL, CT_ALIGN_CORNERS, CT_ASYMMETRIC };
enum { NM_ROUND_PREFER_FLOOR, NM_FLOOR, NM_CEIL, NM_ROUND_PREFER_CEIL };

__device__ __forceinline__ float oe_src_coord(int o, int in, int out, float scale, int ct) {
    switch (ct) {
        case CT_ALIGN_CORNERS: return out > 1 ? tk_divf((float)o * (float)(in - 1), (float)(out - 1)) : 0.0f;
        case CT_ASYMMETRIC: return tk_divf((float)o, scale);
        case CT_PYTORCH_HALF_PIXEL: return out > 1 ? tk_divf((float)o + 0.5f, scale) - 0.5f : 0.0f;
        default: return tk_divf((float)o + 0.5f, scale) - 0.5f;
    }
}
__device__ __forceinline__ int oe_nearest(float x, int in, int nm) {
    float r;
    switch (nm) {
        case NM_FLOOR: r = floorf(x); break;
        case NM_CEIL: r = ceilf(x); break;
        case NM_ROUND_PREFER_CEIL: r = floorf(x + 0.5f); break;
        default: r = ceilf(x - 0.5f); break;
    }
    int i = (int)r;
    return i < 0 ? 0 : i >= in ? in - 1 : i;
}

/* Resize over the last two axes of [NC][H][W]; linear = ONNX's bilinear: the source coordinate is clamped to [0, in - 1], the four
 * neighbours are blended as (1 - dy) ((1 - dx) v00 + dx v01) + dy ((1 - dx) v10 + dx v11) in that order */
__global__ void k_oe_resize(const float* x, float* y, int64_t NC, int H, int W, int Ho, int Wo, int linear, int ct, int nm, float sh, float sw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NC * Ho * Wo) return;
    const int wo = (int)(i % Wo), ho = (int)((i / Wo) % Ho);
    const float* xp = x + (i / ((int64_t)Wo * Ho)) * H * W;
    const float fy = oe_src_coord(ho, H, Ho, sh, ct), fx = oe_src_coord(wo, W, Wo, sw, ct);
    if (!linear) {
        y[i] = xp[(int64_t)oe_nearest(fy, H, nm) * W + oe_nearest(fx, W, nm)];
        return;
    }
    const float cy = tk_fminf(tk_fmaxf(fy, 0.0f), (float)(H - 1)), cx = tk_fminf(tk_fmaxf(fx, 0.0f), (float)(W - 1));
    const int y0 = (int)floorf(cy), x0 = (int)floorf(cx);
    const int y1 = y0 + 1 < H ? y0 + 1 : H - 1, x1 = x0 + 1 < W ? x0 + 1 : W - 1;
    const float dy = cy - (float)y0, dx = cx - (float)x0;
    const float v00 = xp[(int64_t)y0 * W + x0], v01 = xp[(int64_t)y0 * W + x1], v10 = xp[(int64_t)y1 * W + x0], v11 = xp[(int64_t)y1 * W + x1];
    const float top = (1.0f - dx) * v00 + dx * v01, bot = (1.0f - dx) * v10 + dx * v11;
    y[i] = (1.0f - dy) * top + dy * bot;
}

__global__ void k_oe_pool(const float* x, float* y, int64_t NC, int H, int W, int kh, int kw, int sh, int sw, int pt, int pl, int Ho, int Wo, int is_max,
                          int include_pad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NC * Ho * Wo) return;
    const int wo = (int)(i % Wo), ho = (int)((i / Wo) % Ho);
    const float* xp = x + (i / ((int64_t)Wo * Ho)) * H * W;
    float acc = is_max ? -INFINITY : 0.0f;
    int cnt = 0;
    for (int a = 0; a < kh; ++a) {
        const int h = ho * sh - pt + a;
        for (int q = 0; q < kw; ++q) {
            const int w = wo * sw - pl + q;
            const bool in = h >= 0 && h < H && w >= 0 && w < W;
            if (in) {
                const float v = xp[(int64_t)h * W + w];
                acc = is_max ? tk_fmaxf(acc, v) : acc + v;
            }
            if (in || include_pad) ++cnt;
        }
    }
    y[i] = is_max ? acc : tk_divf(acc, (float)(cnt > 0 ? cnt : 1));
}

/* GlobalAveragePool: one workgroup per (n, c) plane; 256 strided partial sums, wave butterflies, ((w0 + w1) + w2) + w3 */
__global__ __launch_bounds__(256) void k_oe_gap(const float* x, float* y, int64_t HW) {
    __shared__ float red[4];
    const float* xp = x + (int64_t)blockIdx.x * HW;
    float s = 0.0f;
    for (int64_t i = threadIdx.x; i < HW; i += 256) s = s + xp[i];
    for (int w = 32; w >= 1; w >>= 1) s = s + __shfl_xor(s, w, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) y[blockIdx.x] = tk_divf(((red[0] + red[1]) + red[2]) + red[3], (float)HW);
}

__global__ void k_oe_batchnorm(const float* x, float* y, const float* scale, const float* bias, const float* mean, const float* var, float eps, int C,
                               int64_t HW, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)((i / HW) % C);
    y[i] = tk_fmaf(tk_divf(x[i] - mean[c], tk_sqrtf(var[c] + eps)), scale[c], bias[c]);
}

/* constant / reflect padding of any axes of a rank <= 4 tensor */
struct OePad { int64_t in[4], out[4], before[4]; };
__global__ void k_oe_pad4(const float* x, float* y, OePad p, int reflect, float cval, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t r = i, off = 0, stride = 1;
    bool inside = true;
    for (int d = 3; d >= 0; --d) {
        int64_t c = r % p.out[d] - p.before[d];
        r /= p.out[d];
        if (c < 0 || c >= p.in[d]) {
            if (reflect) {
                if (c < 0) c = -c;
                if (c >= p.in[d]) c = 2 * (p.in[d] - 1) - c;
                if (c < 0 || c >= p.in[d]) inside = false;
            } else inside = false;
        }
        off += c * stride;
        stride *= p.in[d];
    }
    y[i] = inside ? x[off] : cval;
}

__global__ void k_oe_copy(const float* x, float* y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i];
}

/* ------------------------------------------------------------------ host side ---- */

static dim3 grid_for(int64_t n) { return dim3((unsigned)((n + 127) / 128)); }

void TkOnnxExec::unload() {
    if (!arena_ && consts_.empty() && packed_.empty()) return;
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    for (auto& kv : consts_) if (kv.second.d) (void)hipFree(kv.second.d);
    for (auto& kv : packed_) if (kv.second) (void)hipFree(kv.second);
    if (arena_) (void)hipFree(arena_);
    consts_.clear(); packed_.clear(); vals_.clear();
    arena_ = nullptr;
    stream_ = nullptr;
}

float* TkOnnxExec::alloc(int64_t n) {
    const size_t need = ((size_t)(n > 0 ? n : 1) + 63) & ~(size_t)63;
    if (arena_used_ + need > arena_cap_) return nullptr;
    float* p = arena_ + arena_used_;
    arena_used_ += need;
    if (arena_used_ > arena_peak_) arena_peak_ = arena_used_;
    return p;
}

static const char* kSupported[] = {"Conv", "Relu", "Sigmoid", "Tanh", "Sqrt", "Abs", "Neg", "Exp", "Log", "Add", "Sub", "Mul", "Div", "Pow", "Slice", "Concat",
                                   "Pad", "Transpose", "ReduceMean", "LSTM", "Unsqueeze", "Squeeze", "Reshape", "Flatten", "Identity", "Cast", "Constant",
                                   "Clip", "LeakyRelu", "HardSigmoid", "HardSwish", "Resize", "Upsample", "MaxPool", "AveragePool", "GlobalAveragePool",
                                   "BatchNormalization", "MatMul", "Gemm", "Softmax", "Dropout", "Split",
                                   /* tk_onnx_exec_seq.hip */
                                   "LayerNormalization", "Erf", "Gelu", "Gather", "ReduceSum", "ReduceL2", "ReduceMax", "ReduceMin", "Expand", "Max", "Min", "Where",
                                   "ConvTranspose", "Shape", "If", "Loop", "Scan", "Equal", "Less", "Greater", "Not", "And", "Or"};

bool TkOnnxExec::ops_supported(const TkOnnxGraph& g, std::string* err) {
    std::vector<const TkOnnxNode*> all;
    g.all_nodes(&all); /* the branches of If nodes included */
    for (const TkOnnxNode* ndp : all) {
        const TkOnnxNode& nd = *ndp;
        bool ok = false;
        for (const char* s : kSupported) ok = ok || nd.op == s;
        if (!ok) { *err = "ONNX op '" + nd.op + "' (node '" + nd.name + "') is outside the graph classes this path runs (tk_onnx_exec.h)"; return false; }
        if (nd.op == "If") {
            auto tb = nd.attr.find("then_branch"), eb = nd.attr.find("else_branch");
            if (tb == nd.attr.end() || eb == nd.attr.end() || !tb->second.g || !eb->second.g) { *err = "If node '" + nd.name + "' lacks a then_branch / else_branch graph"; return false; }
            if (tb->second.g->outputs.size() != nd.out.size() || eb->second.g->outputs.size() != nd.out.size()) { *err = "If node '" + nd.name + "': its branches declare another number of outputs than the node"; return false; }
        }
        if (nd.op == "Loop" || nd.op == "Scan") {
            auto bd = nd.attr.find("body");
            if (bd == nd.attr.end() || !bd->second.g) { *err = nd.op + " node '" + nd.name + "' lacks a body graph"; return false; }
            const TkOnnxGraph& b = *bd->second.g;
            const size_t carried = nd.op == "Loop" ? (nd.in.size() >= 2 ? nd.in.size() - 2 : 0) : nd.in.size() - (size_t)std::min<int64_t>(std::max<int64_t>(nd.ai("num_scan_inputs", 0), 0), (int64_t)nd.in.size());
            const bool shape_ok = nd.op == "Loop" ? (nd.in.size() >= 2 && b.inputs.size() == nd.in.size() && b.outputs.size() >= carried + 1 && nd.out.size() == b.outputs.size() - 1)
                                                  : (nd.ai("num_scan_inputs", 0) >= 1 && b.inputs.size() == nd.in.size() && b.outputs.size() >= carried && nd.out.size() == b.outputs.size());
            if (!shape_ok) { *err = nd.op + " node '" + nd.name + "': the body's inputs / outputs do not match the node's"; return false; }
        }
    }
    if (g.outputs.empty()) { *err = "the graph declares no outputs"; return false; }
    return true;
}

bool TkOnnxExec::add_const(const std::string& name, const TkOnnxTensor& t) {
    Val v;
    v.shape = t.dims;
    if (!t.f.empty()) {
        VQ(hipMalloc((void**)&v.d, t.f.size() * 4));
        VQ(hipMemcpy(v.d, t.f.data(), t.f.size() * 4, hipMemcpyHostToDevice));
        if (t.f.size() <= 64) v.host = t.f;
    } else if (!t.i.empty() || t.count() == 0) {
        v.is_int = true;
        v.ints = t.i;
    } else { error = "constant '" + name + "' has a data type this path does not read"; return false; }
    auto old = consts_.find(name);
    if (old != consts_.end() && old->second.d) (void)hipFree(old->second.d);
    consts_[name] = v;
    return true;
}

bool TkOnnxExec::load(const char* path, int device, hipStream_t stream, size_t arena_floats) {
    device_ = device;
    stream_ = stream;
    if (!g_.load(path)) { error = g_.error; return false; }
    if (!ops_supported(g_, &error)) return false;
    VQ(hipSetDevice(device_));
    if (!tk_nn_prepare_device()) { error = "tk_nn_prepare_device failed"; return false; }
    for (const auto& kv : g_.init)
        if (!add_const(kv.first, kv.second)) return false;
    std::vector<const TkOnnxNode*> all;
    g_.all_nodes(&all);
    for (const TkOnnxNode* ndp : all) { /* the initialisers and Constant nodes of If branches live in the same name space (ONNX names are unique per model) */
        const TkOnnxNode& nd = *ndp;
        for (const auto& kv : nd.attr)
            if (kv.second.g)
                for (const auto& iv : kv.second.g->init)
                    if (!add_const(iv.first, iv.second)) return false;
        if (nd.op != "Constant" || nd.out.empty()) continue;
        auto it = nd.attr.find("value");
        if (it == nd.attr.end() || !it->second.has_t) { error = "Constant node without a tensor value"; return false; }
        if (!add_const(nd.out[0], it->second.t)) return false;
    }
    arena_cap_ = arena_floats;
    VQ(hipMalloc((void**)&arena_, arena_cap_ * 4));
    begin();
    return true;
}

void TkOnnxExec::begin() {
    if (!staging_.empty() && stream_) (void)hipStreamSynchronize(stream_); /* an upload of the previous run may still read its host source */
    staging_.clear();
    arena_used_ = 0;
    vals_ = consts_;
}

bool TkOnnxExec::run() {
    for (const auto& nd : g_.nodes) {
        if (nd.op == "Constant") continue;
        if (!exec(nd, vals_)) { if (error.find(nd.op) == std::string::npos) error = nd.op + " (node '" + nd.name + "'): " + error; return false; }
    }
    VQ(hipGetLastError());
    return true;
}

static std::vector<int64_t> strides_of(const std::vector<int64_t>& sh) {
    std::vector<int64_t> st(sh.size(), 1);
    for (int i = (int)sh.size() - 2; i >= 0; --i) st[(size_t)i] = st[(size_t)i + 1] * sh[(size_t)i + 1];
    return st;
}

bool TkOnnxExec::exec(const TkOnnxNode& nd, std::map<std::string, Val>& v) {
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
    {
        bool handled = false;
        bool ok = exec_seq_op(nd, v, &handled);
        if (handled) return ok;
        ok = exec_image_op(nd, v, &handled);
        if (handled) return ok;
    }

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
        if (r > TK_VG_RANK) { error = "rank > 6"; return false; }
        std::vector<int64_t> as(r, 1), bs(r, 1), os(r, 1);
        for (size_t i = 0; i < ac.shape.size(); ++i) as[r - ac.shape.size() + i] = ac.shape[i];
        for (size_t i = 0; i < bc.shape.size(); ++i) bs[r - bc.shape.size() + i] = bc.shape[i];
        for (size_t i = 0; i < r; ++i) {
            if (as[i] != bs[i] && as[i] != 1 && bs[i] != 1) { error = "shapes do not broadcast"; return false; }
            os[i] = std::max(as[i], bs[i]);
        }
        const std::vector<int64_t> sa = strides_of(as), sb = strides_of(bs);
        VgIdx ix{};
        for (int d = 0; d < TK_VG_RANK; ++d) { ix.dim[d] = 1; ix.sa[d] = ix.sb[d] = 0; }
        for (size_t i = 0; i < r; ++i) {
            const size_t d = TK_VG_RANK - r + i;
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
            if (ax < 0 || ax > (int64_t)y.shape.size()) { error = "Flatten axis outside the rank of its input"; return false; }
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
        if (r > TK_VG_RANK) { error = "rank > 6"; return false; }
        std::vector<int64_t> perm;
        if (const auto* p = nd.aints("perm")) perm = *p;
        else for (size_t i = 0; i < r; ++i) perm.push_back((int64_t)(r - 1 - i));
        if (perm.size() != r) { error = "perm has the wrong length"; return false; }
        const std::vector<int64_t> sx = strides_of(xc.shape);
        std::vector<int64_t> os(r);
        VgIdx ix{}, oy{};
        for (int d = 0; d < TK_VG_RANK; ++d) { ix.dim[d] = 1; ix.sa[d] = 0; oy.sa[d] = 0; }
        for (size_t i = 0; i < r; ++i) { if (perm[i] < 0 || perm[i] >= (int64_t)r) { error = "bad perm"; return false; } os[i] = xc.shape[(size_t)perm[i]]; }
        const std::vector<int64_t> so = strides_of(os);
        for (size_t i = 0; i < r; ++i) { const size_t d = TK_VG_RANK - r + i; ix.dim[d] = os[i]; ix.sa[d] = sx[(size_t)perm[i]]; oy.sa[d] = so[i]; }
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
        if (r > TK_VG_RANK) { error = "rank > 6"; return false; }
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
        for (int d = 0; d < TK_VG_RANK; ++d) { ix.dim[d] = 1; ix.sa[d] = 0; oy.sa[d] = 0; }
        int64_t off = 0;
        for (size_t i = 0; i < r; ++i) { const size_t d = TK_VG_RANK - r + i; ix.dim[d] = os[i]; ix.sa[d] = sx[i]; oy.sa[d] = so[i]; off += begin[i] * sx[i]; }
        Val* y = out_f(0, os);
        if (!y) return false;
        if (y->count() > 0) hipLaunchKernelGGL(k_vg_gather, grid_for(y->count()), dim3(128), 0, stream_, xc.d, y->d, ix, off, y->count(), (int64_t)0, oy);
        return true;
    }
    if (op == "Split") { /* equal parts or the sizes in `split` (attribute up to opset 12, second input from 13): one strided copy per output */
        Val* x = need(0, false);
        if (!x) return false;
        const Val xc = *x;
        const size_t r = xc.shape.size();
        if (r == 0 || r > TK_VG_RANK) { error = "rank outside 1..6"; return false; }
        int64_t ax = nd.ai("axis", 0);
        if (ax < 0) ax += (int64_t)r;
        if (ax < 0 || ax >= (int64_t)r) { error = "Split axis outside the rank of its input"; return false; }
        std::vector<int64_t> sizes;
        if (!ints_arg("split", 1, &sizes)) {
            const int64_t n = (int64_t)nd.out.size(), dim = xc.shape[(size_t)ax];
            const int64_t each = (dim + n - 1) / n;
            for (int64_t i = 0; i < n; ++i) sizes.push_back(i + 1 < n ? each : dim - each * (n - 1));
        }
        int64_t tot = 0;
        for (int64_t z : sizes) { if (z < 0) { error = "negative split size"; return false; } tot += z; }
        if (sizes.size() != nd.out.size() || tot != xc.shape[(size_t)ax]) { error = "split sizes do not add up to the axis"; return false; }
        const std::vector<int64_t> si = strides_of(xc.shape);
        int64_t at = 0;
        for (size_t k = 0; k < sizes.size(); ++k) {
            std::vector<int64_t> os = xc.shape;
            os[(size_t)ax] = sizes[k];
            Val* y = out_f(k, os);
            if (!y) return false;
            const std::vector<int64_t> so = strides_of(os);
            VgIdx ix{}, oy{};
            for (int d = 0; d < TK_VG_RANK; ++d) { ix.dim[d] = 1; ix.sa[d] = 0; oy.sa[d] = 0; }
            for (size_t i = 0; i < r; ++i) { const size_t d = TK_VG_RANK - r + i; ix.dim[d] = os[i]; ix.sa[d] = si[i]; oy.sa[d] = so[i]; }
            if (y->count() > 0) hipLaunchKernelGGL(k_vg_gather, grid_for(y->count()), dim3(128), 0, stream_, xc.d, y->d, ix, at * si[(size_t)ax], y->count(), (int64_t)0, oy);
            at += sizes[k];
        }
        return true;
    }
    if (op == "Concat") {
        std::vector<Val> parts;
        for (size_t i = 0; i < nd.in.size(); ++i) { Val* x = need(i, false); if (!x) return false; parts.push_back(*x); }
        if (parts.empty()) { error = "no inputs"; return false; }
        const size_t r = parts[0].shape.size();
        if (r > TK_VG_RANK) { error = "rank > 6"; return false; }
        int64_t ax = nd.ai("axis", 0);
        if (ax < 0) ax += (int64_t)r;
        if (ax < 0 || ax >= (int64_t)r) { error = "Concat axis outside the rank of its inputs"; return false; }
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
            for (int d = 0; d < TK_VG_RANK; ++d) { ix.dim[d] = 1; ix.sa[d] = 0; oy.sa[d] = 0; }
            for (size_t i = 0; i < r; ++i) { const size_t d = TK_VG_RANK - r + i; ix.dim[d] = p.shape[i]; ix.sa[d] = sp[i]; oy.sa[d] = so[i]; }
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
        if (r == 0) { error = "Pad of a rank-0 tensor"; return false; }
        if (pads.size() != 2 * r) { error = "pads have the wrong length"; return false; }
        const std::string mode = nd.as("mode", "constant");
        if (mode != "constant" && mode != "reflect") { error = "pad mode '" + mode + "' is not supported"; return false; }
        float cval = nd.af("value", 0.0f);
        if (Val* cv = in(2)) { if (!cv->is_int && !cv->host.empty()) cval = cv->host[0]; }
        bool only_last = true;
        for (size_t i = 0; i + 1 < r; ++i) only_last = only_last && pads[i] == 0 && pads[r + i] == 0;
        if (!only_last) {
            if (r > 4) { error = "rank > 4"; return false; }
            OePad pp{};
            for (int d = 0; d < 4; ++d) { pp.in[d] = pp.out[d] = 1; pp.before[d] = 0; }
            std::vector<int64_t> os(r);
            for (size_t i = 0; i < r; ++i) {
                const size_t d = 4 - r + i;
                if (pads[i] < 0 || pads[r + i] < 0 || (mode == "reflect" && (pads[i] >= xc.shape[i] || pads[r + i] >= xc.shape[i]))) { error = "bad pad amounts"; return false; }
                pp.in[d] = xc.shape[i];
                pp.before[d] = pads[i];
                pp.out[d] = os[i] = xc.shape[i] + pads[i] + pads[r + i];
            }
            Val* y = out_f(0, os);
            if (!y) return false;
            hipLaunchKernelGGL(k_oe_pad4, grid_for(y->count()), dim3(128), 0, stream_, xc.d, y->d, pp, mode == "reflect" ? 1 : 0, cval, y->count());
            return true;
        }
        const int64_t L = xc.shape[r - 1], pb = pads[r - 1], pe = pads[2 * r - 1], Lo = L + pb + pe;
        if (pb < 0 || pe < 0 || (mode == "reflect" && (pb >= L || pe >= L))) { error = "bad pad amounts"; return false; }
        std::vector<int64_t> os = xc.shape;
        os[r - 1] = Lo;
        Val* y = out_f(0, os);
        if (!y) return false;
        const int64_t rows = xc.count() / (L > 0 ? L : 1);
        hipLaunchKernelGGL(k_vg_pad_last, grid_for(y->count()), dim3(128), 0, stream_, xc.d, y->d, rows, L, pb, Lo, mode == "reflect" ? 1 : 0, cval);
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

/* ------------------------------------------------------------------ the ops of the 2-D image graphs ---- */

bool TkOnnxExec::exec_image_op(const TkOnnxNode& nd, std::map<std::string, Val>& v, bool* handled) {
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
    auto pair_attr = [&](const char* k, int dflt, int* a, int* b) {
        *a = *b = dflt;
        if (const auto* p = nd.aints(k)) if (p->size() == 2) { *a = (int)(*p)[0]; *b = (int)(*p)[1]; }
    };
    const std::string& op = nd.op;
    *handled = true;

    if (op == "Dropout") { /* inference: identity */
        Val* x = in(0);
        if (!x) { error = "input 0 is missing"; return false; }
        v[nd.out[0]] = *x;
        return true;
    }
    if (op == "Clip" || op == "LeakyRelu" || op == "HardSigmoid" || op == "HardSwish") {
        Val* x = need(0);
        if (!x) return false;
        const Val xc = *x;
        float p0 = 0.0f, p1 = 0.0f;
        int pop;
        if (op == "Clip") {
            pop = P_CLIP;
            p0 = nd.has("min") ? nd.af("min", 0.0f) : -INFINITY;
            p1 = nd.has("max") ? nd.af("max", 0.0f) : INFINITY;
            if (Val* lo = in(1)) { if (lo->host.empty()) { error = "the lower bound must be a constant scalar"; return false; } p0 = lo->host[0]; }
            if (Val* hi = in(2)) { if (hi->host.empty()) { error = "the upper bound must be a constant scalar"; return false; } p1 = hi->host[0]; }
        } else if (op == "LeakyRelu") { pop = P_LEAKY; p0 = nd.af("alpha", 0.01f); }
        else if (op == "HardSigmoid") { pop = P_HSIGMOID; p0 = nd.af("alpha", 0.2f); p1 = nd.af("beta", 0.5f); }
        else pop = P_HSWISH;
        Val* y = out_f(0, xc.shape);
        if (!y) return false;
        hipLaunchKernelGGL(k_oe_unary_p, grid_for(xc.count()), dim3(128), 0, stream_, pop, xc.d, y->d, xc.count(), p0, p1);
        return true;
    }
    if (op == "Conv") {
        Val* x = in(0);
        if (!x || x->shape.size() != 4) { *handled = false; return true; } /* 1-D signal convolutions: the caller's path */
        Val* w = need(1);
        if (!w || x->is_int) { if (x && x->is_int) error = "input 0 must be a float tensor"; return false; }
        const Val xc = *x, wc = *w;
        Val* b = in(2);
        if (wc.shape.size() != 4) { error = "a 4-D input needs [M, C / group, kh, kw] weights"; return false; }
        OeConv p{};
        p.groups = (int)nd.ai("group", 1);
        const int N = (int)xc.shape[0];
        p.C = (int)xc.shape[1]; p.H = (int)xc.shape[2]; p.W = (int)xc.shape[3];
        p.M = (int)wc.shape[0]; p.kh = (int)wc.shape[2]; p.kw = (int)wc.shape[3];
        if (p.groups < 1 || p.C % p.groups || p.M % p.groups || wc.shape[1] != p.C / p.groups) { error = "channel counts and group do not fit"; return false; }
        pair_attr("strides", 1, &p.sh, &p.sw);
        pair_attr("dilations", 1, &p.dh, &p.dw);
        int pb = 0, pr = 0;
        if (const auto* pd = nd.aints("pads")) if (pd->size() == 4) { p.pt = (int)(*pd)[0]; p.pl = (int)(*pd)[1]; pb = (int)(*pd)[2]; pr = (int)(*pd)[3]; }
        const std::string ap = nd.as("auto_pad", "NOTSET");
        if (ap == "SAME_UPPER" || ap == "SAME_LOWER") {
            const int oh = (p.H + p.sh - 1) / p.sh, ow = (p.W + p.sw - 1) / p.sw;
            const int th = std::max(0, (oh - 1) * p.sh + (p.kh - 1) * p.dh + 1 - p.H), tw = std::max(0, (ow - 1) * p.sw + (p.kw - 1) * p.dw + 1 - p.W);
            p.pt = ap == "SAME_UPPER" ? th / 2 : th - th / 2; pb = th - p.pt;
            p.pl = ap == "SAME_UPPER" ? tw / 2 : tw - tw / 2; pr = tw - p.pl;
        } else if (ap != "NOTSET" && ap != "VALID") { error = "auto_pad '" + ap + "' is not supported"; return false; }
        if (p.sh < 1 || p.sw < 1 || p.dh < 1 || p.dw < 1) { error = "bad strides / dilations"; return false; }
        p.Ho = (p.H + p.pt + pb - p.dh * (p.kh - 1) - 1) / p.sh + 1;
        p.Wo = (p.W + p.pl + pr - p.dw * (p.kw - 1) - 1) / p.sw + 1;
        if (p.Ho < 1 || p.Wo < 1) { error = "empty output"; return false; }
        if (b && (b->is_int || b->count() != p.M)) { error = "bias has the wrong size"; return false; }
        const float* bd = b ? b->d : nullptr;
        Val* y = out_f(0, {N, p.M, p.Ho, p.Wo});
        if (!y) return false;
        if (p.groups > 1) {
            hipLaunchKernelGGL(k_oe_conv2d_direct, grid_for(y->count()), dim3(128), 0, stream_, xc.d, wc.d, bd, y->d, p, N);
            return true;
        }
        /* dense: y_n [M][Ho Wo] = Wp [M][K1] x col_n [Ho Wo][K1]^T on the exact fp32 MFMA GEMM; the bias is column K of Wp against a column of ones */
        const int K = p.C * p.kh * p.kw, K1 = (K + 1 + 3) & ~3;
        const std::string key = nd.name.empty() ? nd.out[0] : nd.name + "/" + nd.out[0];
        float* wp = nullptr;
        auto pk = packed_.find(key);
        if (pk != packed_.end()) wp = pk->second;
        else {
            if (consts_.find(nd.in[1]) == consts_.end() || (b && consts_.find(nd.in[2]) == consts_.end())) { error = "weights and bias must be constants"; return false; }
            VQ(hipMalloc((void**)&wp, (size_t)p.M * K1 * 4));
            hipLaunchKernelGGL(k_oe_pack_conv, grid_for((int64_t)p.M * K1), dim3(128), 0, stream_, wc.d, bd, wp, p.M, K, K1);
            packed_[key] = wp;
        }
        const int64_t HW = (int64_t)p.Ho * p.Wo;
        const size_t mark = arena_used_; /* the column matrix is scratch: its space is handed back below (reuse is stream-ordered) */
        float* col = alloc(HW * K1);
        if (!col) { error = "activation arena exhausted (im2col)"; return false; }
        for (int n = 0; n < N; ++n) { /* one image at a time: the column matrix is the largest buffer of the run */
            const int64_t total = HW * K1;
            const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 65535 * 16);
            hipLaunchKernelGGL(k_oe_im2col, dim3(blocks), dim3(256), 0, stream_, xc.d + (int64_t)n * p.C * p.H * p.W, p, col, K, K1);
            TkGemm g{};
            g.A = wp; g.B = col; g.C = y->d + (int64_t)n * p.M * HW;
            g.M = p.M; g.N = (int)HW; g.K = K1; g.lda = K1; g.ldb = K1; g.ldc = (int)HW;
            g.b_kn = 0; g.act = TK_ACT_NONE; g.alpha = 1.0f; g.batch = 1;
            tk_launch_gemm(g, stream_);
        }
        arena_used_ = mark;
        return true;
    }
    if (op == "Resize" || op == "Upsample") {
        Val* x = need(0);
        if (!x) return false;
        const Val xc = *x;
        if (xc.shape.size() != 4) { error = "expects [N, C, H, W]"; return false; }
        const std::string mode = nd.as("mode", "nearest");
        if (mode != "nearest" && mode != "linear") { error = "mode '" + mode + "' is not supported"; return false; }
        const int H = (int)xc.shape[2], W = (int)xc.shape[3];
        int Ho = 0, Wo = 0;
        float sh = 0.0f, sw = 0.0f;
        std::vector<float> scales;
        if (op == "Upsample") {
            if (const auto* sc = nd.afloats("scales")) scales = *sc;
            else if (Val* s = in(1)) scales = s->host;
        } else {
            Val* s = in(2);
            if (s && !s->host.empty()) scales = s->host;
            Val* sz = in(3);
            if (sz && sz->is_int && sz->ints.size() == 4) {
                if (sz->ints[0] != xc.shape[0] || sz->ints[1] != xc.shape[1]) { error = "only the spatial axes can be resized"; return false; }
                Ho = (int)sz->ints[2]; Wo = (int)sz->ints[3];
                sh = (float)Ho / (float)H; sw = (float)Wo / (float)W;
            }
            if (nd.in.size() == 2) { Val* s1 = in(1); if (s1 && !s1->host.empty()) scales = s1->host; } /* opset 10: X, scales */
        }
        if (Ho == 0) {
            if (scales.size() != 4 || scales[0] != 1.0f || scales[1] != 1.0f) { error = "needs constant scales [1, 1, sh, sw] or sizes"; return false; }
            sh = scales[2]; sw = scales[3];
            Ho = (int)floorf((float)H * sh); Wo = (int)floorf((float)W * sw);
        }
        if (Ho < 1 || Wo < 1) { error = "empty output"; return false; }
        const std::string cts = op == "Upsample" ? "asymmetric" : nd.as("coordinate_transformation_mode", "half_pixel");
        const int ct = cts == "half_pixel" ? CT_HALF_PIXEL : cts == "pytorch_half_pixel" ? CT_PYTORCH_HALF_PIXEL : cts == "align_corners" ? CT_ALIGN_CORNERS
                       : cts == "asymmetric" ? CT_ASYMMETRIC : -1;
        if (ct < 0) { error = "coordinate_transformation_mode '" + cts + "' is not supported"; return false; }
        const std::string nms = op == "Upsample" ? "floor" : nd.as("nearest_mode", "round_prefer_floor");
        const int nm = nms == "floor" ? NM_FLOOR : nms == "ceil" ? NM_CEIL : nms == "round_prefer_ceil" ? NM_ROUND_PREFER_CEIL : NM_ROUND_PREFER_FLOOR;
        Val* y = out_f(0, {xc.shape[0], xc.shape[1], Ho, Wo});
        if (!y) return false;
        hipLaunchKernelGGL(k_oe_resize, grid_for(y->count()), dim3(128), 0, stream_, xc.d, y->d, xc.shape[0] * xc.shape[1], H, W, Ho, Wo, mode == "linear" ? 1 : 0, ct,
                           nm, sh, sw);
        return true;
    }
    if (op == "MaxPool" || op == "AveragePool" || op == "GlobalAveragePool") {
        Val* x = need(0);
        if (!x) return false;
        const Val xc = *x;
        if (xc.shape.size() != 4) { error = "expects [N, C, H, W]"; return false; }
        const int H = (int)xc.shape[2], W = (int)xc.shape[3];
        int kh = H, kw = W, sh = 1, sw = 1, pt = 0, pl = 0, pb = 0, pr = 0;
        if (op != "GlobalAveragePool") {
            const auto* ks = nd.aints("kernel_shape");
            if (!ks || ks->size() != 2) { error = "needs a 2-D kernel_shape"; return false; }
            kh = (int)(*ks)[0]; kw = (int)(*ks)[1];
            pair_attr("strides", 1, &sh, &sw);
            if (const auto* pd = nd.aints("pads")) if (pd->size() == 4) { pt = (int)(*pd)[0]; pl = (int)(*pd)[1]; pb = (int)(*pd)[2]; pr = (int)(*pd)[3]; }
            if (nd.as("auto_pad", "NOTSET") != "NOTSET") { error = "auto_pad is not supported"; return false; }
            if (const auto* dl = nd.aints("dilations")) for (int64_t d : *dl) if (d != 1) { error = "dilated pooling is not supported"; return false; }
        }
        const bool ceil_mode = nd.ai("ceil_mode", 0) != 0;
        auto odim = [&](int in_, int k, int s, int pa, int pz) {
            const int num = in_ + pa + pz - k;
            int o = (ceil_mode ? (num + s - 1) / s : num / s) + 1;
            if (ceil_mode && (o - 1) * s >= in_ + pa) --o; /* the last window must start inside the input or its leading pad */
            return o;
        };
        const int Ho = odim(H, kh, sh, pt, pb), Wo = odim(W, kw, sw, pl, pr);
        if (kh < 1 || kw < 1 || sh < 1 || sw < 1 || Ho < 1 || Wo < 1) { error = "empty output"; return false; }
        Val* y = out_f(0, {xc.shape[0], xc.shape[1], Ho, Wo});
        if (!y) return false;
        if (op == "GlobalAveragePool") {
            hipLaunchKernelGGL(k_oe_gap, dim3((unsigned)(xc.shape[0] * xc.shape[1])), dim3(256), 0, stream_, xc.d, y->d, (int64_t)H * W);
            return true;
        }
        hipLaunchKernelGGL(k_oe_pool, grid_for(y->count()), dim3(128), 0, stream_, xc.d, y->d, xc.shape[0] * xc.shape[1], H, W, kh, kw, sh, sw, pt, pl, Ho, Wo,
                           op == "MaxPool" ? 1 : 0, (int)nd.ai("count_include_pad", 0));
        return true;
    }
    if (op == "BatchNormalization") {
        Val* x = need(0);
        Val *sc = x ? need(1) : nullptr, *bi = sc ? need(2) : nullptr, *mu = bi ? need(3) : nullptr, *va = mu ? need(4) : nullptr;
        if (!va) return false;
        const Val xc = *x;
        if (xc.shape.size() < 2) { error = "expects [N, C, ...]"; return false; }
        const int C = (int)xc.shape[1];
        if (sc->count() != C || bi->count() != C || mu->count() != C || va->count() != C) { error = "per-channel parameters have the wrong size"; return false; }
        int64_t HW = 1;
        for (size_t i = 2; i < xc.shape.size(); ++i) HW *= xc.shape[i];
        const float *scd = sc->d, *bid = bi->d, *mud = mu->d, *vad = va->d;
        Val* y = out_f(0, xc.shape);
        if (!y) return false;
        hipLaunchKernelGGL(k_oe_batchnorm, grid_for(xc.count()), dim3(128), 0, stream_, xc.d, y->d, scd, bid, mud, vad, nd.af("epsilon", 1e-5f), C, HW, xc.count());
        return true;
    }
    if (op == "MatMul" || op == "Gemm") {
        Val* a = need(0);
        Val* b = a ? need(1) : nullptr;
        if (!b) return false;
        const Val ac = *a, bc = *b;
        const bool gemm = op == "Gemm";
        const bool tb = gemm && nd.ai("transB", 0) != 0;
        if (gemm && (nd.ai("transA", 0) != 0 || nd.af("alpha", 1.0f) != 1.0f || nd.af("beta", 1.0f) != 1.0f)) { error = "only alpha = beta = 1, transA = 0"; return false; }
        if (ac.shape.size() < 2 || bc.shape.size() != 2 || (gemm && ac.shape.size() != 2)) { error = "expects A [..., M, K] and a 2-D B"; return false; }
        const int K = (int)ac.shape.back(), N = (int)(tb ? bc.shape[0] : bc.shape[1]);
        if ((tb ? bc.shape[1] : bc.shape[0]) != K) { error = "inner dimensions differ"; return false; }
        const int64_t M = ac.count() / K;
        const float* bias = nullptr;
        if (gemm) if (Val* c = in(2)) { if (c->is_int || c->count() != N) { error = "C must hold N values"; return false; } bias = c->d; }
        std::vector<int64_t> os = ac.shape;
        os.back() = N;
        Val* y = out_f(0, os);
        if (!y) return false;
        TkGemm g{};
        g.A = ac.d; g.B = bc.d; g.C = y->d; g.bias = bias;
        g.M = (int)M; g.N = N; g.K = K; g.lda = K; g.ldb = tb ? K : N; g.ldc = N;
        g.b_kn = tb ? 0 : 1; g.act = TK_ACT_NONE; g.alpha = 1.0f; g.batch = 1;
        tk_launch_gemm(g, stream_);
        return true;
    }
    if (op == "Softmax") {
        Val* x = need(0);
        if (!x) return false;
        const Val xc = *x;
        const int64_t r = (int64_t)xc.shape.size();
        int64_t ax = nd.ai("axis", -1);
        if (ax < 0) ax += r;
        if (r < 1 || ax < 0 || ax >= r) { error = "Softmax axis outside the rank of its input"; return false; }
        if (ax != r - 1) { /* an inner axis (the DFL of a YOLO head: [1, 16, 4, anchors] over axis 1): one thread per (outer, inner) pair walks the axis in index order */
            Val* y = out_f(0, xc.shape);
            if (!y) return false;
            int64_t inner = 1;
            for (int64_t i = ax + 1; i < r; ++i) inner *= xc.shape[(size_t)i];
            const int64_t n = xc.shape[(size_t)ax], pairs = n > 0 ? xc.count() / n : 0;
            if (pairs > 0) hipLaunchKernelGGL(k_oe_softmax_axis, grid_for(pairs), dim3(128), 0, stream_, xc.d, y->d, pairs, n, inner);
            return true;
        }
        Val* y = out_f(0, xc.shape);
        if (!y) return false;
        const int cols = (int)xc.shape.back();
        hipLaunchKernelGGL(k_oe_copy, grid_for(xc.count()), dim3(128), 0, stream_, xc.d, y->d, xc.count());
        tk_launch_softmax_rows(y->d, (int)(xc.count() / cols), cols, cols, stream_);
        return true;
    }
    *handled = false;
    return true;
}
