#include "tk_vision_engine.h"
#include "tk_onnx_weights.h"

#include <stdio.h>
#include <string.h>

#include "../nn/tk_nn_kernels.h"

/* Perception streams run at the highest stream priority: their kernels are small and many, and behind the LLM's large GEMM
 * launches they would otherwise queue for a free CU at every step (the fused cycle's critical path becomes queueing, not work). */
static inline hipError_t tk_create_perception_stream(hipStream_t* s) {
    int lo = 0, hi = 0; /* numerically lower = higher priority */
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi);
}

#define HIPQ(expr)                                                                              \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess) {                                                                \
            char b__[256];                                                                      \
            snprintf(b__, sizeof b__, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            error = b__;                                                                        \
            return false;                                                                       \
        }                                                                                       \
    } while (0)

/* ------------------------------------------------------------------------------------------
 * pre-processing: the reference CPU formula, op for op (scalar branch):
 *   ratio = (orig - 1) / target ; g = ratio * o ; i = (int)g ; d = g - i
 *   v = p1*(1-dx)*(1-dy) + p2*dx*(1-dy) + p3*(1-dx)*dy + p4*dx*dy      (left to right)
 *   out = (v / 255 - mean) / std
 * compiled with -ffp-contract=off; bit-identical to the compiled reference object (tests).
 * One thread per output pixel, all three channels; u8 reads are served by L2 (the 4 taps of
 * neighbouring pixels overlap), fp32 writes are coalesced per plane.  HBM-bound: 1.23 MB in, 4.92 MB out.
 * ------------------------------------------------------------------------------------------ */
__global__ void k_preprocess(TkPreprocessArgs a) {
    const uint32_t ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y * blockDim.y + threadIdx.y;
    if (ox >= a.out_w || oy >= a.out_h) return;
    const float x_ratio = tk_divf((float)a.in_w - 1.0f, (float)a.out_w);
    const float y_ratio = tk_divf((float)a.in_h - 1.0f, (float)a.out_h);
    const float gx = x_ratio * (float)ox, gy = y_ratio * (float)oy;
    const int x = (int)gx, y = (int)gy;
    const float xd = gx - (float)x, yd = gy - (float)y;
    const int x1 = x + 1 < (int)a.in_w ? x + 1 : x, y1 = y + 1 < (int)a.in_h ? y + 1 : y; /* reference reads x+1/y+1 (always in range for w,h >= 2) */
    const uint8_t* r0 = a.src + (size_t)y * a.in_stride;
    const uint8_t* r1 = a.src + (size_t)y1 * a.in_stride;
    const size_t np = (size_t)a.out_w * a.out_h, pix = (size_t)oy * a.out_w + ox;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float p1 = (float)r0[x * a.bpp + c], p2 = (float)r0[x1 * a.bpp + c];
        const float p3 = (float)r1[x * a.bpp + c], p4 = (float)r1[x1 * a.bpp + c];
        float v = (p1 * (1.0f - xd)) * (1.0f - yd);
        v = v + (p2 * xd) * (1.0f - yd);
        v = v + (p3 * (1.0f - xd)) * yd;
        v = v + (p4 * xd) * yd;
        const float o = tk_divf(tk_divf(v, 255.0f) - a.mean[c], a.std_dev[c]);
        if (a.nhwc) a.dst[pix * 3 + c] = o;
        else a.dst[c * np + pix] = o;
    }
}

void tk_launch_preprocess(const TkPreprocessArgs& a, hipStream_t s) {
    dim3 block(64, 4);
    dim3 grid((a.out_w + 63) / 64, (a.out_h + 3) / 4);
    hipLaunchKernelGGL(k_preprocess, grid, block, 0, s, a);
}

/* ------------------------------------------------------------------------------------------ model */

TkYoloModel::~TkYoloModel() {
    (void)hipSetDevice(device);
    for (auto p : w) if (p) (void)hipFree(p);
    for (auto p : b) if (p) (void)hipFree(p);
}

bool TkYoloModel::init(int dev, int n_classes) {
    device = dev;
    nc = n_classes;
    if (nc < 1 || nc > 1024) { error = "class count out of range"; return false; }
    struct Dummy {};
    specs = TkYoloV8n<Dummy>::specs(nc);
    HIPQ(hipSetDevice(device));
    w.assign(specs.size(), nullptr);
    b.assign(specs.size(), nullptr);
    for (size_t i = 0; i < specs.size(); ++i) {
        const TkConvSpec& s = specs[i];
        HIPQ(hipMalloc((void**)&w[i], (size_t)s.cout * s.k * s.k * s.cin * 4));
        HIPQ(hipMalloc((void**)&b[i], (size_t)s.cout * 4));
    }
    return true;
}

size_t TkYoloModel::param_count() const {
    size_t n = 0;
    for (const auto& s : specs) n += (size_t)s.cout * s.k * s.k * s.cin + s.cout;
    return n;
}

bool TkYoloModel::set_layer(int idx, const float* w_host, const float* b_host) {
    if (idx < 0 || idx >= (int)specs.size()) { error = "layer index out of range"; return false; }
    const TkConvSpec& s = specs[idx];
    HIPQ(hipSetDevice(device));
    HIPQ(hipMemcpy(w[idx], w_host, (size_t)s.cout * s.k * s.k * s.cin * 4, hipMemcpyHostToDevice));
    HIPQ(hipMemcpy(b[idx], b_host, (size_t)s.cout * 4, hipMemcpyHostToDevice));
    return true;
}

bool TkYoloModel::fill_synthetic(uint64_t seed, float cls_bias) {
    std::vector<float> hw, hb;
    const int n = (int)specs.size();
    for (int i = 0; i < n; ++i) {
        const TkConvSpec& s = specs[i];
        const int fan = s.k * s.k * s.cin;
        hw.resize((size_t)s.cout * fan);
        hb.resize(s.cout);
        /* the last class conv of each scale: layers n-13, n-7, n-1 (6 convs per scale, class 1x1 last) */
        const bool cls_out = (i == n - 1 || i == n - 7 || i == n - 13);
        for (size_t j = 0; j < hw.size(); ++j) hw[j] = tk_yolo_synth_w(seed, i, (int64_t)j, fan);
        for (int j = 0; j < s.cout; ++j) hb[j] = tk_yolo_synth_b(seed, i, j, cls_out, cls_bias);
        if (!set_layer(i, hw.data(), hb.data())) return false;
    }
    return true;
}

bool TkYoloModel::load_file(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { error = std::string("cannot open ") + path; return false; }
    char magic[8];
    int32_t hdr[2];
    bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "TKYOLO1", 8) == 0 && fread(hdr, 4, 2, f) == 2 && hdr[0] == (int)specs.size() && hdr[1] == nc;
    std::vector<float> hw, hb;
    for (size_t i = 0; ok && i < specs.size(); ++i) {
        const TkConvSpec& s = specs[i];
        int32_t d[4];
        ok = fread(d, 4, 4, f) == 4 && d[0] == s.cin && d[1] == s.cout && d[2] == s.k && d[3] == s.s;
        hw.resize((size_t)s.cout * s.k * s.k * s.cin);
        hb.resize(s.cout);
        ok = ok && fread(hw.data(), 4, hw.size(), f) == hw.size() && fread(hb.data(), 4, hb.size(), f) == hb.size();
        ok = ok && set_layer((int)i, hw.data(), hb.data());
    }
    fclose(f);
    if (!ok && error.empty()) error = "not a TKYOLO1 container for this geometry";
    return ok;
}

/* a YOLO-class file that is not the YOLOv8n topology: accepted when its graph is one the executor can run — every op supported, one float image
 * input [N, 3, H, W], a first output [N, 4 + nc, anchors] (checked against the engine's geometry at the first run) */
static bool yolo_generic_graph(const char* path, TkYoloModel* m, const std::string& why_not_v8n) {
    TkOnnxGraph g;
    if (!g.load(path)) { m->error = why_not_v8n + "; and as a graph: " + g.error; return false; }
    std::string err;
    if (!TkOnnxExec::ops_supported(g, &err)) { m->error = why_not_v8n + "; and the graph executor cannot run it: " + err; return false; }
    std::string in;
    for (const auto& vi : g.inputs) {
        if (g.init.count(vi.name) == 0 && (vi.elem_type == 1 || vi.elem_type == 0)) { /* (the reader already leaves initialisers out of `inputs`) */ if (!in.empty()) { m->error = why_not_v8n + "; and the graph has more than one float input"; return false; } in = vi.name; }
    }
    if (in.empty() || g.outputs.empty()) { m->error = why_not_v8n + "; and the graph has no float image input / no output"; return false; }
    m->generic = true;
    m->onnx_path = path;
    m->graph_in = in;
    m->graph_out = g.outputs[0].name;
    return true;
}

bool TkYoloModel::load_onnx(const char* path) {
    TkOnnxWeights ox;
    /* the Conv-initialiser reader only understands what a YOLOv8n export holds; a file it cannot read may still be a graph the executor runs */
    if (!ox.load(path)) return yolo_generic_graph(path, this, "as a YOLOv8n weight file: " + ox.error);
    /* 63 graph convolutions, optionally followed by the DFL projection conv [1][16][1][1] */
    size_t n = ox.convs.size();
    if (n == specs.size() + 1 && ox.convs.back().cout == 1 && ox.convs.back().cin == TK_YOLO_REG_MAX && ox.convs.back().kh == 1) --n;
    if (n != specs.size())
        return yolo_generic_graph(path, this, "the ONNX graph has " + std::to_string(ox.convs.size()) + " Conv nodes, YOLOv8n has " + std::to_string(specs.size()) + " (+ DFL)");
    for (size_t i = 0; i < n; ++i) { /* the hard-wired graph is the fast path: only a file that IS that topology takes it */
        const TkConvSpec& s = specs[i];
        const TkOnnxConv& c = ox.convs[i];
        if (c.cout != s.cout || c.cin != s.cin || c.kh != s.k || c.kw != s.k)
            return yolo_generic_graph(path, this, "Conv " + std::to_string(i) + " is [" + std::to_string(c.cout) + "][" + std::to_string(c.cin) + "][" + std::to_string(c.kh) + "][" +
                                                      std::to_string(c.kw) + "], the YOLOv8n graph expects [" + std::to_string(s.cout) + "][" + std::to_string(s.cin) + "][" +
                                                      std::to_string(s.k) + "][" + std::to_string(s.k) + "]");
    }
    std::vector<float> hw;
    for (size_t i = 0; i < n; ++i) {
        const TkConvSpec& s = specs[i];
        const TkOnnxConv& c = ox.convs[i];
        if (c.cout != s.cout || c.cin != s.cin || c.kh != s.k || c.kw != s.k) {
            error = "Conv " + std::to_string(i) + " does not match the YOLOv8n graph";
            return false;
        }
        hw.resize(c.w.size());
        for (int o = 0; o < s.cout; ++o) /* [cout][cin][kh][kw] -> [cout][kh][kw][cin] */
            for (int ci = 0; ci < s.cin; ++ci)
                for (int k = 0; k < s.k * s.k; ++k) hw[((size_t)o * s.k * s.k + k) * s.cin + ci] = c.w[((size_t)o * s.cin + ci) * s.k * s.k + k];
        if (!set_layer((int)i, hw.data(), c.b.data())) return false;
    }
    return true;
}

/* ------------------------------------------------------------------------------------------ graph ops on the GPU */

struct TkGpuOps {
    TkDetector* d;
    hipStream_t s;
    TkT alloc(int B, int H, int W, int C) {
        TkT t;
        t.B = B; t.H = H; t.W = W; t.C = C; t.ld = C;
        size_t n = ((size_t)B * H * W * C + 63) & ~(size_t)63;
        t.p = d->arena + d->arena_used;
        d->arena_used += n;
        return t;
    }
    void conv(const TkT& x, int idx, const TkT& y, const TkT* res) {
        const TkConvSpec& sp = d->model->specs[idx];
        TkGemm g{};
        g.M = y.B * y.H * y.W; g.N = sp.cout; g.K = sp.k * sp.k * sp.cin;
        if (sp.k == 1 && sp.s == 1) { g.A = x.p; g.lda = x.ld; }
        else if (tk_gemm_im2col_ok(x.p, x.C, x.ld, d->model->w[idx])) { /* the GEMM addresses the input directly: no column matrix */
            g.A = x.p; g.lda = 0;
            g.im_C = x.C; g.im_H = x.H; g.im_W = x.W; g.im_ldx = x.ld; g.im_kw = sp.k; g.im_stride = sp.s; g.im_pad = sp.k / 2;
            g.im_Ho = y.H; g.im_Wo = y.W;
        } else if (!res && tk_launch_conv_stem(x.p, x.B, x.H, x.W, x.C, x.ld, d->model->w[idx], d->model->b[idx], sp.act ? TK_ACT_SILU : TK_ACT_NONE, sp.cout, sp.k,
                                               sp.s, sp.k / 2, y.p, y.ld, s)) {
            return; /* the 3-channel stem: a direct kernel, the GEMM's chain */
        } else {
            tk_launch_im2col(x.p, x.B, x.H, x.W, x.C, x.ld, sp.k, sp.k, sp.s, sp.k / 2, d->col, s);
            g.A = d->col; g.lda = g.K;
        }
        g.B = d->model->w[idx]; g.ldb = g.K; g.b_kn = 0;
        g.C = y.p; g.ldc = y.ld;
        g.bias = d->model->b[idx];
        g.residual = res ? res->p : nullptr; g.ldr = res ? res->ld : 0;
        g.act = sp.act ? TK_ACT_SILU : TK_ACT_NONE;
        g.alpha = 1.0f; g.batch = 1;
        g.fast = d->fast ? 1 : 0;
        tk_launch_gemm(g, s);
    }
    void maxpool5(const TkT& x, const TkT& y) { tk_launch_maxpool5(x.p, x.B, x.H, x.W, x.C, x.ld, y.p, y.ld, s); }
    void upsample2x(const TkT& x, const TkT& y) { tk_launch_upsample2x(x.p, x.B, x.H, x.W, x.C, x.ld, y.p, y.ld, s); }
    void copy(const TkT& x, const TkT& y) { tk_launch_copy_cols(x.p, x.B * x.H * x.W, x.C, x.ld, y.p, y.ld, s); }
};

/* sizing pass: same graph, counts arena floats and the largest im2col matrix */
struct TkSizeOps {
    const std::vector<TkConvSpec>* specs;
    size_t used = 0, col = 0;
    TkT alloc(int B, int H, int W, int C) {
        TkT t;
        t.B = B; t.H = H; t.W = W; t.C = C; t.ld = C;
        used += ((size_t)B * H * W * C + 63) & ~(size_t)63;
        return t;
    }
    void conv(const TkT&, int idx, const TkT& y, const TkT*) {
        const TkConvSpec& sp = (*specs)[idx];
        if (!(sp.k == 1 && sp.s == 1)) { size_t n = (size_t)y.B * y.H * y.W * sp.k * sp.k * sp.cin; col = n > col ? n : col; }
    }
    void maxpool5(const TkT&, const TkT&) {}
    void upsample2x(const TkT&, const TkT&) {}
    void copy(const TkT&, const TkT&) {}
};

/* ------------------------------------------------------------------------------------------ decode + NMS kernels */

__global__ void k_yolo_decode(TkT h0, TkT h1, TkT h2, int nc, int n_anchors, float conf, tk_yolo_cand_t* cand) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (a >= n_anchors) return;
    const int n0 = h0.H * h0.W, n1 = h1.H * h1.W;
    const TkT* h;
    int local;
    float stride;
    if (a < n0) { h = &h0; local = a; stride = 8.0f; }
    else if (a < n0 + n1) { h = &h1; local = a - n0; stride = 16.0f; }
    else { h = &h2; local = a - n0 - n1; stride = 32.0f; }
    const float ax = (float)(local % h->W) + 0.5f, ay = (float)(local / h->W) + 0.5f;
    tk_yolo_cand_t c;
    tk_yolo_decode_anchor(h->p + ((size_t)b * h->H * h->W + local) * h->ld, nc, ax, ay, stride, &c);
    c.anchor = a;
    if (!(c.score > conf)) c.cls = -1; /* not a candidate */
    cand[(size_t)b * n_anchors + a] = c;
}

/* a YOLO-class graph's own output [4 + nc][anchors] (Ultralytics export: box centre / size in input pixels, class probabilities already through
 * the sigmoid): xyxy = centre -/+ size / 2, score = max_c p_c (first class on ties), candidate iff score > confidence_threshold */
__global__ void k_yolo_decode_out(const float* out, int nc, int n_anchors, float conf, tk_yolo_cand_t* cand) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_anchors) return;
    tk_yolo_cand_t c;
    tk_yolo_decode_out_anchor(out + a, n_anchors, nc, &c);
    c.anchor = a;
    if (!(c.score > conf)) c.cls = -1;
    cand[a] = c;
}

/* rank of every candidate among the candidates: (score desc, anchor asc); the best MAX_CAND are kept.  Most anchors fall below the
 * confidence threshold, so each 256-anchor tile is compacted in LDS first (ballot + prefix) and only its survivors are compared. */
__global__ __launch_bounds__(256) void k_yolo_rank(const tk_yolo_cand_t* cand, int n_anchors, int32_t* order, int32_t* n_cand) {
    const int b = blockIdx.y;
    const tk_yolo_cand_t* cb = cand + (size_t)b * n_anchors;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float ss[256];
    __shared__ int sj[256];
    __shared__ int wcnt[4];
    const bool valid = a < n_anchors && cb[a].cls >= 0;
    const float mys = valid ? cb[a].score : 0.0f;
    int rank = 0, total = 0;
    for (int base = 0; base < n_anchors; base += 256) {
        const int j = base + threadIdx.x;
        const bool v = j < n_anchors && cb[j].cls >= 0;
        const float sco = v ? cb[j].score : 0.0f;
        const uint64_t bal = __ballot(v);
        if (lane == 0) wcnt[wave] = __popcll(bal);
        __syncthreads();
        int off = 0, nt = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int c = wcnt[w]; if (w < wave) off += c; nt += c; }
        if (v) {
            const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
            ss[pos] = sco; sj[pos] = j;
        }
        __syncthreads();
        if (valid)
            for (int k = 0; k < nt; ++k)
                if (ss[k] > mys || (ss[k] == mys && sj[k] < a)) ++rank;
        total += nt;
        __syncthreads();
    }
    if (valid && rank < TK_YOLO_MAX_CAND) order[(size_t)b * TK_YOLO_MAX_CAND + rank] = a;
    if (a == 0) n_cand[b] = total < TK_YOLO_MAX_CAND ? total : TK_YOLO_MAX_CAND;
}

/* suppression bit matrix: bit j of mask[i] set iff j > i, same class, IoU > thr */
__global__ void k_yolo_mask(const tk_yolo_cand_t* cand, int n_anchors, const int32_t* order, const int32_t* n_cand, float iou_thr, uint64_t* mask) {
    const int b = blockIdx.z;
    const int n = n_cand[b];
    const int i = blockIdx.y * blockDim.y + threadIdx.y;
    const int wj = blockIdx.x * blockDim.x + threadIdx.x; /* 64-candidate word */
    if (i >= n || wj * 64 >= n) return;
    const tk_yolo_cand_t* cb = cand + (size_t)b * n_anchors;
    const int32_t* ob = order + (size_t)b * TK_YOLO_MAX_CAND;
    const tk_yolo_cand_t ci = cb[ob[i]];
    uint64_t bits = 0;
    for (int k = 0; k < 64; ++k) {
        const int j = wj * 64 + k;
        if (j <= i || j >= n) continue;
        const tk_yolo_cand_t cj = cb[ob[j]];
        if (cj.cls == ci.cls && tk_yolo_iou(&ci, &cj) > iou_thr) bits |= 1ull << k;
    }
    mask[((size_t)b * TK_YOLO_MAX_CAND + i) * (TK_YOLO_MAX_CAND / 64) + wj] = bits;
}

/* greedy scan, one wave per frame */
__global__ void k_yolo_scan(const tk_yolo_cand_t* cand, int n_anchors, const int32_t* order, const int32_t* n_cand, const uint64_t* mask,
                            TkDetection* kept, int32_t* n_kept) {
    const int b = blockIdx.x, lane = threadIdx.x;
    __shared__ uint64_t removed[TK_YOLO_MAX_CAND / 64];
    const int n = n_cand[b];
    const int words = (n + 63) / 64;
    for (int w = lane; w < TK_YOLO_MAX_CAND / 64; w += 64) removed[w] = 0;
    __syncthreads();
    const tk_yolo_cand_t* cb = cand + (size_t)b * n_anchors;
    const int32_t* ob = order + (size_t)b * TK_YOLO_MAX_CAND;
    int nk = 0;
    for (int i = 0; i < n && nk < TK_OBJECT_DETECTOR_MAX_DETECTIONS; ++i) {
        const bool dead = (removed[i >> 6] >> (i & 63)) & 1;
        if (!dead) {
            const uint64_t* mrow = mask + ((size_t)b * TK_YOLO_MAX_CAND + i) * (TK_YOLO_MAX_CAND / 64);
            for (int w = lane; w < words; w += 64) removed[w] |= mrow[w];
            if (lane == 0) {
                const tk_yolo_cand_t c = cb[ob[i]];
                TkDetection dt{c.x1, c.y1, c.x2, c.y2, c.score, c.cls, c.anchor};
                kept[(size_t)b * TK_OBJECT_DETECTOR_MAX_DETECTIONS + nk] = dt;
            }
            ++nk;
        }
        __syncthreads();
    }
    if (lane == 0) n_kept[b] = nk;
}

/* ------------------------------------------------------------------------------------------ detector */

TkDetector::~TkDetector() {
    if (model) (void)hipSetDevice(model->device);
    if (stream) (void)hipStreamSynchronize(stream);
    void* ptrs[] = {frame_dev, arena, col, input, cand, order, n_cand, mask, kept, n_kept, attr_dev};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (exec) exec->unload(); /* before its stream goes */
    if (stream) (void)hipStreamDestroy(stream);
}

bool TkDetector::init(TkYoloModel* m, int w, int h, int mb) {
    model = m; in_w = w; in_h = h; max_batch = mb;
    if (!m || w <= 0 || h <= 0 || (w % 32) || (h % 32)) { error = "detector input must be a positive multiple of 32"; return false; }
    if (mb < 1 || mb > 256) { error = "max_batch must be in [1,256]"; return false; }
    HIPQ(hipSetDevice(m->device));
    HIPQ(tk_create_perception_stream(&stream));
    n_anchors = (h / 8) * (w / 8) + (h / 16) * (w / 16) + (h / 32) * (w / 32);
    if (m->generic) {
        /* the file's own graph, one frame per run: every node's output of a run lives in the executor's arena (a YOLO-nano-class graph at 640 x 640
         * holds ~25 M activation floats + im2col scratch; 320 floats per input pixel leaves a wide margin and is checked by the executor) */
        exec.reset(new TkOnnxExec());
        if (!exec->load(m->onnx_path.c_str(), m->device, stream, (size_t)320 * h * w + (1u << 22))) { error = "detector graph: " + exec->error; return false; }
    } else {
        TkSizeOps so{&m->specs};
        TkT x = so.alloc(mb, h, w, 3);
        so.used = 0;
        TkT o[3];
        TkYoloV8n<TkSizeOps>::forward(so, x, o, m->nc);
        arena_floats = so.used + 1024;
        col_floats = so.col + 1024;
        HIPQ(hipMalloc((void**)&arena, arena_floats * 4));
        HIPQ(hipMalloc((void**)&col, col_floats * 4));
    }
    HIPQ(hipMalloc((void**)&input, (size_t)mb * h * w * 3 * 4));
    HIPQ(hipMalloc((void**)&cand, (size_t)mb * n_anchors * sizeof(tk_yolo_cand_t)));
    HIPQ(hipMalloc((void**)&order, (size_t)mb * TK_YOLO_MAX_CAND * 4));
    HIPQ(hipMalloc((void**)&n_cand, (size_t)mb * 4));
    HIPQ(hipMalloc((void**)&mask, (size_t)mb * TK_YOLO_MAX_CAND * (TK_YOLO_MAX_CAND / 64) * 8));
    HIPQ(hipMalloc((void**)&kept, (size_t)mb * TK_OBJECT_DETECTOR_MAX_DETECTIONS * sizeof(TkDetection)));
    HIPQ(hipMalloc((void**)&n_kept, (size_t)mb * 4));
    return true;
}

/* generic models: frame by frame through the file's graph (exports carry a batch dimension of one), each frame's output decoded before the next
 * run reuses the arena (same stream: in order) */
bool TkDetector::run_graph(int B, float* host_out) {
    for (int b = 0; b < B; ++b) {
        exec->begin();
        TkOnnxExec::Val in;
        in.d = input + (size_t)b * 3 * in_h * in_w;
        in.shape = {1, 3, in_h, in_w};
        exec->bind(model->graph_in, in);
        if (!exec->run()) { error = "detector graph: " + exec->error; return false; }
        const TkOnnxExec::Val* o = exec->value(model->graph_out);
        if (!o || !o->d || o->is_int) { error = "detector graph: no float output"; return false; }
        /* [1, 4 + nc, anchors] (or [4 + nc, anchors]) with this engine's anchor count */
        const size_t r = o->shape.size();
        if (r < 2 || o->shape[r - 1] != n_anchors || o->shape[r - 2] != 4 + model->nc || o->count() != (int64_t)(4 + model->nc) * n_anchors) {
            std::string sh;
            for (int64_t d : o->shape) sh += (sh.empty() ? "" : ", ") + std::to_string(d);
            error = "detector graph: output [" + sh + "], expected [1, " + std::to_string(4 + model->nc) + ", " + std::to_string(n_anchors) + "] (4 + class_count rows, " +
                    std::to_string(in_w) + " x " + std::to_string(in_h) + " at strides 8 / 16 / 32)";
            return false;
        }
        hipLaunchKernelGGL(k_yolo_decode_out, dim3((n_anchors + 255) / 256), dim3(256), 0, stream, o->d, model->nc, n_anchors, conf, cand + (size_t)b * n_anchors);
        if (host_out) HIPQ(hipMemcpyAsync(host_out + (size_t)b * o->count(), o->d, (size_t)o->count() * 4, hipMemcpyDeviceToHost, stream)); /* before the next run reuses the arena */
    }
    return true;
}

bool TkDetector::forward_graph(int B, const float* nchw_host, std::vector<float>* out) {
    if (!model->generic) { error = "forward_graph is the graph-executor path's hook (this model runs the hard-wired YOLOv8n graph: forward_tensor)"; return false; }
    if (B < 1 || B > max_batch) { error = "batch larger than the detector was created for"; return false; }
    HIPQ(hipSetDevice(model->device));
    HIPQ(hipMemcpyAsync(input, nchw_host, (size_t)B * 3 * in_h * in_w * 4, hipMemcpyHostToDevice, stream));
    out->assign((size_t)B * (4 + model->nc) * n_anchors, 0.0f);
    last_B = 0;
    if (!run_graph(B, out->data())) return false;
    hipLaunchKernelGGL(k_yolo_rank, dim3((n_anchors + 255) / 256, B), dim3(256), 0, stream, cand, n_anchors, order, n_cand);
    hipLaunchKernelGGL(k_yolo_mask, dim3(TK_YOLO_MAX_CAND / 64 / 8, TK_YOLO_MAX_CAND / 8, B), dim3(8, 8), 0, stream, cand, n_anchors, order, n_cand, iou, mask);
    hipLaunchKernelGGL(k_yolo_scan, dim3(B), dim3(64), 0, stream, cand, n_anchors, order, n_cand, mask, kept, n_kept);
    HIPQ(hipGetLastError());
    HIPQ(hipStreamSynchronize(stream));
    return true;
}

bool TkDetector::enqueue(int B) {
    if (model->generic) {
        if (!run_graph(B, nullptr)) return false;
        hipLaunchKernelGGL(k_yolo_rank, dim3((n_anchors + 255) / 256, B), dim3(256), 0, stream, cand, n_anchors, order, n_cand);
        hipLaunchKernelGGL(k_yolo_mask, dim3(TK_YOLO_MAX_CAND / 64 / 8, TK_YOLO_MAX_CAND / 8, B), dim3(8, 8), 0, stream, cand, n_anchors, order, n_cand, iou, mask);
        hipLaunchKernelGGL(k_yolo_scan, dim3(B), dim3(64), 0, stream, cand, n_anchors, order, n_cand, mask, kept, n_kept);
        HIPQ(hipGetLastError());
        return true;
    }
    TkGpuOps ops{this, stream};
    arena_used = 0;
    TkT x;
    x.p = input; x.B = B; x.H = in_h; x.W = in_w; x.C = 3; x.ld = 3;
    TkYoloV8n<TkGpuOps>::forward(ops, x, heads, model->nc);
    hipLaunchKernelGGL(k_yolo_decode, dim3((n_anchors + 255) / 256, B), dim3(256), 0, stream, heads[0], heads[1], heads[2], model->nc, n_anchors, conf, cand);
    hipLaunchKernelGGL(k_yolo_rank, dim3((n_anchors + 255) / 256, B), dim3(256), 0, stream, cand, n_anchors, order, n_cand);
    hipLaunchKernelGGL(k_yolo_mask, dim3(TK_YOLO_MAX_CAND / 64 / 8, TK_YOLO_MAX_CAND / 8, B), dim3(8, 8), 0, stream, cand, n_anchors, order, n_cand, iou, mask);
    hipLaunchKernelGGL(k_yolo_scan, dim3(B), dim3(64), 0, stream, cand, n_anchors, order, n_cand, mask, kept, n_kept);
    HIPQ(hipGetLastError());
    return true;
}

bool TkDetector::detect(int B, const uint8_t* const* frames, uint32_t w, uint32_t h, uint32_t stride, uint32_t bpp, std::vector<std::vector<TkDetection>>* out) {
    if (B < 1 || B > max_batch) { error = "batch larger than the detector was created for"; return false; }
    if (w < 2 || h < 2 || stride < w * bpp) { error = "frame geometry invalid (need w,h >= 2 and stride >= w*bpp)"; return false; }
    HIPQ(hipSetDevice(model->device));
    const size_t fb = (size_t)stride * h;
    if (fb * B > frame_cap) { /* once per frame geometry: room for the engine's whole batch, so a wider job later never re-allocates (hipFree synchronises the device) */
        if (frame_dev) (void)hipFree(frame_dev);
        frame_dev = nullptr;
        HIPQ(hipMalloc((void**)&frame_dev, fb * max_batch));
        frame_cap = fb * max_batch;
    }
    for (int b = 0; b < B; ++b) {
        HIPQ(hipMemcpyAsync(frame_dev + fb * b, frames[b], fb, hipMemcpyHostToDevice, stream));
        TkPreprocessArgs a{};
        a.src = frame_dev + fb * b; a.in_w = w; a.in_h = h; a.in_stride = stride; a.bpp = bpp;
        a.dst = input + (size_t)b * in_h * in_w * 3; a.out_w = in_w; a.out_h = in_h; a.nhwc = model->generic ? 0 : 1; /* a file's graph takes planar NCHW */
        for (int c = 0; c < 3; ++c) { a.mean[c] = mean[c]; a.std_dev[c] = std_dev[c]; }
        tk_launch_preprocess(a, stream);
    }
    last_w = w; last_h = h; last_stride = stride; last_B = B;
    if (!enqueue(B)) return false;
    return fetch(B, out);
}

/* ------------------------------------------------------------------------------------------
 * Per-box attributes (reference: src/vision/tk_attribute_classifier.c, run per detection by tk_vision_pipeline.c:462-485).
 * One workgroup per box; both classifiers are byte / integer histograms over the box's pixels, so the counts are exact whatever
 * the order the threads visit them in, and the fp32 / fp64 arithmetic of the HSV test is the reference's, operation for operation:
 *   r,g,b = u8 / 255.0f;  v = max, delta = max - min, s = delta / max  (s = 0 when max = 0)
 *   h = 60 * ((g - b) / delta | 2 + (b - r) / delta | 4 + (r - g) / delta), + 360 when negative   (the constants are doubles in the
 *   reference: a float op double op rounds once more on the store to float, reproduced with explicit double arithmetic)
 *   s < 0.1 (double compare): black (v < 0.1) / white (v > 0.9) / gray, else six 60-degree hue bins starting at 330.
 * Dominant colour = first bin with the maximum count.  Like the reference the frame is read as tightly packed RGB8
 * (row pitch = 3 * width, :56); pixels outside the frame are skipped (:54).
 * Door state: count interior pixels whose luminance (integer 299/587/114 per mille) differs by more than 100 between the rows above
 * and below; closed when count / (w * h) > 0.1.  The reference does not bounds-check this loop (:113-117); out-of-frame pixels are
 * skipped here.
 * ------------------------------------------------------------------------------------------ */
__device__ __forceinline__ int tk_color_bin(uint8_t R, uint8_t G, uint8_t B) {
    const float r = tk_divf((float)R, 255.0f), g = tk_divf((float)G, 255.0f), b = tk_divf((float)B, 255.0f);
    const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
    const float delta = mx - mn, v = mx;
    float s = 0.0f, h = 0.0f;
    if ((double)mx > 0.0) {
        s = tk_divf(delta, mx);
        if (r >= mx) h = tk_divf(g - b, delta);
        else if (g >= mx) h = (float)(2.0 + (double)tk_divf(b - r, delta));
        else h = (float)(4.0 + (double)tk_divf(r - g, delta));
        h = (float)((double)h * 60.0);
        if ((double)h < 0.0) h = (float)((double)h + 360.0);
    }
    if ((double)s < 0.1) return (double)v < 0.1 ? 6 : ((double)v > 0.9 ? 7 : 8);
    if (h < 30.0f || h >= 330.0f) return 0;
    if (h < 90.0f) return 1;
    if (h < 150.0f) return 2;
    if (h < 210.0f) return 3;
    if (h < 270.0f) return 4;
    if (h < 330.0f) return 5;
    return -1; /* NaN hue: the reference counts the pixel in no bin */
}

__device__ __forceinline__ int tk_luma(const uint8_t* p) { return (int)(uint8_t)((p[0] * 299 + p[1] * 587 + p[2] * 114) / 1000); }

__global__ __launch_bounds__(256) void k_box_attributes(const uint8_t* frame, int W, int H, const int32_t* rects, int32_t* out) {
    __shared__ int bins[9];
    __shared__ int edges;
    const int box = blockIdx.x, t = threadIdx.x;
    const int bx = rects[4 * box], by = rects[4 * box + 1], bw = rects[4 * box + 2], bh = rects[4 * box + 3];
    if (t < 9) bins[t] = 0;
    if (t == 0) edges = 0;
    __syncthreads();
    const int64_t npix = (bw > 0 && bh > 0) ? (int64_t)bw * bh : 0;
    for (int64_t i = t; i < npix; i += 256) {
        const int x = bx + (int)(i % bw), y = by + (int)(i / bw);
        if (x < 0 || x >= W || y < 0 || y >= H) continue;
        const uint8_t* p = frame + ((int64_t)y * W + x) * 3;
        const int c = tk_color_bin(p[0], p[1], p[2]);
        if (c >= 0) atomicAdd(&bins[c], 1);
        const int lx = x - bx, ly = y - by; /* door-state loop: interior of the box, rows above / below inside the frame */
        if (lx >= 1 && lx < bw - 1 && ly >= 1 && ly < bh - 1 && y >= 1 && y < H - 1) {
            const int d = tk_luma(p - (int64_t)W * 3) - tk_luma(p + (int64_t)W * 3);
            if ((d < 0 ? -d : d) > 100) atomicAdd(&edges, 1);
        }
    }
    __syncthreads();
    if (t == 0) {
        int best = 0;
        for (int i = 1; i < 9; ++i)
            if (bins[i] > bins[best]) best = i;
        out[2 * box] = best;
        const float density = tk_divf((float)edges, (float)(bw * bh));
        out[2 * box + 1] = (double)density > 0.1 ? 1 : 0;
    }
}

bool TkDetector::run_attributes(const uint8_t* frame, uint32_t w, uint32_t h, int n, const int32_t* rects, int32_t* color, int32_t* door_closed) {
    if (n <= 0) return true;
    if (n > attr_cap) { /* at least the detector's own cap of boxes at the first use: no regrowth afterwards */
        const int want = n > TK_OBJECT_DETECTOR_MAX_DETECTIONS ? n : TK_OBJECT_DETECTOR_MAX_DETECTIONS;
        if (attr_dev) (void)hipFree(attr_dev);
        attr_dev = nullptr;
        attr_cap = 0;
        HIPQ(hipMalloc((void**)&attr_dev, (size_t)want * 6 * 4));
        attr_cap = want;
    }
    int32_t* res = attr_dev + (size_t)attr_cap * 4;
    HIPQ(hipMemcpyAsync(attr_dev, rects, (size_t)n * 16, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_box_attributes, dim3(n), dim3(256), 0, stream, frame, (int)w, (int)h, attr_dev, res);
    std::vector<int32_t> hr((size_t)n * 2);
    HIPQ(hipMemcpyAsync(hr.data(), res, (size_t)n * 8, hipMemcpyDeviceToHost, stream));
    HIPQ(hipStreamSynchronize(stream));
    for (int i = 0; i < n; ++i) {
        if (color) color[i] = hr[2 * i];
        if (door_closed) door_closed[i] = hr[2 * i + 1];
    }
    return true;
}

bool TkDetector::classify_boxes(int b, int n, const int32_t* rects, int32_t* color, int32_t* door_closed) {
    if (b < 0 || b >= last_B || !frame_dev) { error = "no frame of that index is resident (call detect first)"; return false; }
    if (last_stride != last_w * 3) { error = "attribute classification reads tightly packed RGB8 frames, as the reference does"; return false; }
    HIPQ(hipSetDevice(model->device));
    return run_attributes(frame_dev + (size_t)last_stride * last_h * b, last_w, last_h, n, rects, color, door_closed);
}

bool tk_classify_boxes_host(int device, const uint8_t* frame, uint32_t w, uint32_t h, int n, const int32_t* rects, int32_t* color,
                            int32_t* door_closed, std::string* error) {
    if (n <= 0) return true;
    uint8_t* df = nullptr;
    int32_t* dr = nullptr;
    const size_t fb = (size_t)w * h * 3;
    bool ok = hipSetDevice(device) == hipSuccess && hipMalloc((void**)&df, fb) == hipSuccess && hipMalloc((void**)&dr, (size_t)n * 6 * 4) == hipSuccess;
    std::vector<int32_t> hr((size_t)n * 2);
    if (ok) ok = hipMemcpy(df, frame, fb, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dr, rects, (size_t)n * 16, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_box_attributes, dim3(n), dim3(256), 0, nullptr, df, (int)w, (int)h, dr, dr + (size_t)n * 4);
        ok = hipGetLastError() == hipSuccess && hipMemcpy(hr.data(), dr + (size_t)n * 4, (size_t)n * 8, hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (df) (void)hipFree(df);
    if (dr) (void)hipFree(dr);
    if (!ok) { if (error) *error = "HIP error in the attribute classifier"; return false; }
    for (int i = 0; i < n; ++i) {
        if (color) color[i] = hr[2 * i];
        if (door_closed) door_closed[i] = hr[2 * i + 1];
    }
    return true;
}

bool TkDetector::fetch(int B, std::vector<std::vector<TkDetection>>* out) {
    std::vector<int32_t> nk(B);
    std::vector<TkDetection> hk((size_t)B * TK_OBJECT_DETECTOR_MAX_DETECTIONS);
    HIPQ(hipMemcpyAsync(nk.data(), n_kept, B * 4, hipMemcpyDeviceToHost, stream));
    HIPQ(hipMemcpyAsync(hk.data(), kept, hk.size() * sizeof(TkDetection), hipMemcpyDeviceToHost, stream));
    HIPQ(hipStreamSynchronize(stream));
    out->assign(B, {});
    for (int b = 0; b < B; ++b)
        (*out)[b].assign(hk.begin() + (size_t)b * TK_OBJECT_DETECTOR_MAX_DETECTIONS, hk.begin() + (size_t)b * TK_OBJECT_DETECTOR_MAX_DETECTIONS + nk[b]);
    return true;
}

bool TkDetector::forward_tensor(int B, const float* nhwc_host, std::vector<float>* raw_out) {
    if (model->generic) { error = "raw head maps exist on the hard-wired YOLOv8n path only; this model runs its file's graph (forward_graph)"; return false; }
    if (B < 1 || B > max_batch) { error = "batch larger than the detector was created for"; return false; }
    HIPQ(hipSetDevice(model->device));
    HIPQ(hipMemcpyAsync(input, nhwc_host, (size_t)B * in_h * in_w * 3 * 4, hipMemcpyHostToDevice, stream));
    if (!enqueue(B)) return false;
    const int no = 64 + model->nc;
    raw_out->resize((size_t)B * n_anchors * no);
    size_t off = 0;
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < 3; ++i) {
            const size_t n = (size_t)heads[i].H * heads[i].W * no;
            HIPQ(hipMemcpyAsync(raw_out->data() + off, heads[i].p + (size_t)b * n, n * 4, hipMemcpyDeviceToHost, stream));
            off += n;
        }
    HIPQ(hipStreamSynchronize(stream));
    return true;
}
