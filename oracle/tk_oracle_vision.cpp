/*
 * tk_oracle_vision.cpp — TEST INFRASTRUCTURE ONLY (oracle) for the detector stream.
 *
 * Pinned parts:
 *   orc_preprocess  — restates src/vision/tk_image_preprocessor.c:43-69 (bilinear stretch, ratio (orig-1)/target,
 *                     no half-pixel offset) and :156-160 (scalar normalisation x/255, -mean, /std).  Pinned bit-for-bit
 *                     against the COMPILED reference function (oracle/_ref/libtkref_preprocess.so, tests/golden/preprocess_*.npz).
 * PARITY UNPINNED parts (the arithmetic lives in ONNX Runtime, un-pinned and absent — SURVEY.md §0 F1/F3/F6):
 *   YOLOv8n forward — the graph of trackiellm_amd/csrc/common/tk_yolov8n_graph.h walked with scalar CPU ops; pinned against an
 *                     independent torch implementation (tests/golden/make_vision_golden.py -> tests/golden/yolo_tiny.npz).
 *   post-processing — restates what src/vision/tk_object_detector.h:118-143 promises (threshold, NMS, original-frame
 *                     coordinates); the reference body is a stub (tk_object_detector.c:303-368).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../trackiellm_amd/csrc/common/tk_ggml_blocks.h"
#include "../trackiellm_amd/csrc/common/tk_yolo_post.h"
#include "../trackiellm_amd/csrc/common/tk_yolov8n_graph.h"

extern "C" {

int orc_preprocess(const uint8_t* data, uint32_t w, uint32_t h, uint32_t stride, uint32_t bpp, float* out, uint32_t tw, uint32_t th,
                   const float* mean, const float* std_dev, int nhwc) {
    if (!data || !out || tw == 0 || th == 0) return 1001;
    const float x_ratio = ((float)w - 1.0f) / (float)tw;
    const float y_ratio = ((float)h - 1.0f) / (float)th;
    const size_t np = (size_t)tw * th;
    for (uint32_t oy = 0; oy < th; ++oy)
        for (uint32_t ox = 0; ox < tw; ++ox) {
            const float gx = x_ratio * (float)ox, gy = y_ratio * (float)oy;
            const int x = (int)gx, y = (int)gy;
            const float xd = gx - (float)x, yd = gy - (float)y;
            const int x1 = x + 1 < (int)w ? x + 1 : x, y1 = y + 1 < (int)h ? y + 1 : y;
            for (int c = 0; c < 3; ++c) {
                const float p1 = data[(size_t)y * stride + x * bpp + c], p2 = data[(size_t)y * stride + x1 * bpp + c];
                const float p3 = data[(size_t)y1 * stride + x * bpp + c], p4 = data[(size_t)y1 * stride + x1 * bpp + c];
                float v = (p1 * (1.0f - xd)) * (1.0f - yd);
                v = v + (p2 * xd) * (1.0f - yd);
                v = v + (p3 * (1.0f - xd)) * yd;
                v = v + (p4 * xd) * yd;
                const float o = (v / 255.0f - mean[c]) / std_dev[c];
                if (nhwc) out[((size_t)oy * tw + ox) * 3 + c] = o;
                else out[c * np + (size_t)oy * tw + ox] = o;
            }
        }
    return 0;
}

struct orc_yolo {
    int nc;
    std::vector<TkConvSpec> specs;
    std::vector<std::vector<float>> w, b;
};

struct CpuOps {
    orc_yolo* m;
    std::vector<std::vector<float>> bufs;
    TkT alloc(int B, int H, int W, int C) {
        bufs.emplace_back((size_t)B * H * W * C, 0.0f);
        TkT t;
        t.p = bufs.back().data(); t.B = B; t.H = H; t.W = W; t.C = C; t.ld = C;
        return t;
    }
    void conv(const TkT& x, int idx, const TkT& y, const TkT* res) {
        const TkConvSpec& sp = m->specs[idx];
        const int k = sp.k, s = sp.s, pad = k / 2, C = sp.cin;
        const float* W = m->w[idx].data();
        const float* bias = m->b[idx].data();
        const int64_t npix = (int64_t)y.B * y.H * y.W;
#pragma omp parallel for schedule(static)
        for (int64_t pix = 0; pix < npix; ++pix) {
            const int ox = (int)(pix % y.W), oy = (int)((pix / y.W) % y.H), b = (int)(pix / ((int64_t)y.W * y.H));
            std::vector<float> patch((size_t)k * k * C);
            for (int ky = 0; ky < k; ++ky)
                for (int kx = 0; kx < k; ++kx) {
                    const int iy = oy * s + ky - pad, ix = ox * s + kx - pad;
                    const bool in = iy >= 0 && iy < x.H && ix >= 0 && ix < x.W;
                    const float* src = in ? x.p + (((int64_t)b * x.H + iy) * x.W + ix) * x.ld : nullptr;
                    for (int c = 0; c < C; ++c) patch[((size_t)ky * k + kx) * C + c] = in ? src[c] : 0.0f;
                }
            const int K = k * k * C;
            for (int co = 0; co < sp.cout; ++co) {
                const float* wr = W + (size_t)co * K;
                float acc = 0.0f;
                for (int kk = 0; kk < K; ++kk) acc = tk_fmaf(patch[kk], wr[kk], acc); /* the MFMA's k-ordered chain */
                float v = acc + bias[co];
                if (sp.act) v = tk_siluf(v);
                if (res) v = v + res->p[pix * res->ld + co];
                y.p[pix * y.ld + co] = v;
            }
        }
    }
    void maxpool5(const TkT& x, const TkT& y) {
        for (int b = 0; b < x.B; ++b)
            for (int oy = 0; oy < x.H; ++oy)
                for (int ox = 0; ox < x.W; ++ox)
                    for (int c = 0; c < x.C; ++c) {
                        float mx = -INFINITY;
                        for (int dy = -2; dy <= 2; ++dy)
                            for (int dx = -2; dx <= 2; ++dx) {
                                const int iy = oy + dy, ix = ox + dx;
                                if (iy >= 0 && iy < x.H && ix >= 0 && ix < x.W) mx = tk_fmaxf(mx, x.p[(((int64_t)b * x.H + iy) * x.W + ix) * x.ld + c]);
                            }
                        y.p[(((int64_t)b * x.H + oy) * x.W + ox) * y.ld + c] = mx;
                    }
    }
    void upsample2x(const TkT& x, const TkT& y) {
        for (int b = 0; b < x.B; ++b)
            for (int oy = 0; oy < 2 * x.H; ++oy)
                for (int ox = 0; ox < 2 * x.W; ++ox)
                    for (int c = 0; c < x.C; ++c)
                        y.p[(((int64_t)b * 2 * x.H + oy) * 2 * x.W + ox) * y.ld + c] = x.p[(((int64_t)b * x.H + oy / 2) * x.W + ox / 2) * x.ld + c];
    }
    void copy(const TkT& x, const TkT& y) {
        const int64_t n = (int64_t)x.B * x.H * x.W;
        for (int64_t i = 0; i < n; ++i)
            for (int c = 0; c < x.C; ++c) y.p[i * y.ld + c] = x.p[i * x.ld + c];
    }
};

orc_yolo* orc_yolo_create(int nc, uint64_t seed, float cls_bias) {
    orc_yolo* m = new orc_yolo();
    m->nc = nc;
    m->specs = TkYoloV8n<CpuOps>::specs(nc);
    const int n = (int)m->specs.size();
    m->w.resize(n);
    m->b.resize(n);
    for (int i = 0; i < n; ++i) {
        const TkConvSpec& s = m->specs[i];
        const int fan = s.k * s.k * s.cin;
        m->w[i].resize((size_t)s.cout * fan);
        m->b[i].resize(s.cout);
        const bool cls_out = (i == n - 1 || i == n - 7 || i == n - 13);
        for (size_t j = 0; j < m->w[i].size(); ++j) m->w[i][j] = tk_yolo_synth_w(seed, i, (int64_t)j, fan);
        for (int j = 0; j < s.cout; ++j) m->b[i][j] = tk_yolo_synth_b(seed, i, j, cls_out, cls_bias);
    }
    return m;
}
void orc_yolo_destroy(orc_yolo* m) { delete m; }
int orc_yolo_layer_count(orc_yolo* m) { return (int)m->specs.size(); }
void orc_yolo_layer_spec(orc_yolo* m, int i, int32_t* out5) {
    const TkConvSpec& s = m->specs[i];
    out5[0] = s.cin; out5[1] = s.cout; out5[2] = s.k; out5[3] = s.s; out5[4] = s.act;
}
/* weights in the product's order: [cout][(ky*k + kx)*cin + c] */
void orc_yolo_get_layer(orc_yolo* m, int i, float* w, float* b) {
    memcpy(w, m->w[i].data(), m->w[i].size() * 4);
    memcpy(b, m->b[i].data(), m->b[i].size() * 4);
}

/* in: [B][H][W][3] pre-processed; raw: [B][anchors][64+nc] (scales 8, 16, 32 concatenated) */
void orc_yolo_forward(orc_yolo* m, int B, int H, int W, const float* in, float* raw) {
    CpuOps ops{m};
    ops.bufs.reserve(512);
    TkT x;
    x.p = const_cast<float*>(in); x.B = B; x.H = H; x.W = W; x.C = 3; x.ld = 3;
    TkT o[3];
    TkYoloV8n<CpuOps>::forward(ops, x, o, m->nc);
    const int no = 64 + m->nc;
    size_t off = 0;
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < 3; ++i) {
            const size_t n = (size_t)o[i].H * o[i].W * no;
            memcpy(raw + off, o[i].p + (size_t)b * n, n * 4);
            off += n;
        }
}

/* raw: [anchors][64+nc] of ONE frame -> survivors; returns count; boxes5 = x1,y1,x2,y2,score (input-tensor pixels) */
int orc_yolo_post(const float* raw, int H, int W, int nc, float conf, float iou, float* boxes5, int32_t* cls, int32_t* anchors, int cap) {
    const int no = 64 + nc;
    std::vector<tk_yolo_cand_t> cand;
    int a = 0;
    const int strides[3] = {8, 16, 32};
    for (int i = 0; i < 3; ++i) {
        const int gh = H / strides[i], gw = W / strides[i];
        for (int l = 0; l < gh * gw; ++l, ++a) {
            tk_yolo_cand_t c;
            tk_yolo_decode_anchor(raw + (size_t)a * no, nc, (float)(l % gw) + 0.5f, (float)(l / gw) + 0.5f, (float)strides[i], &c);
            c.anchor = a;
            if (c.score > conf) cand.push_back(c);
        }
    }
    std::stable_sort(cand.begin(), cand.end(), [](const tk_yolo_cand_t& p, const tk_yolo_cand_t& q) {
        return p.score > q.score || (p.score == q.score && p.anchor < q.anchor);
    });
    if ((int)cand.size() > TK_YOLO_MAX_CAND) cand.resize(TK_YOLO_MAX_CAND);
    std::vector<tk_yolo_cand_t> kept;
    for (const auto& c : cand) {
        bool drop = false;
        for (const auto& k : kept)
            if (k.cls == c.cls && tk_yolo_iou(&k, &c) > iou) { drop = true; break; }
        if (!drop) {
            kept.push_back(c);
            if ((int)kept.size() >= TK_OBJECT_DETECTOR_MAX_DETECTIONS) break;
        }
    }
    for (int i = 0; i < (int)kept.size() && i < cap; ++i) {
        boxes5[5 * i] = kept[i].x1; boxes5[5 * i + 1] = kept[i].y1; boxes5[5 * i + 2] = kept[i].x2; boxes5[5 * i + 3] = kept[i].y2;
        boxes5[5 * i + 4] = kept[i].score;
        cls[i] = kept[i].cls;
        anchors[i] = kept[i].anchor;
    }
    return (int)kept.size();
}

/* out: [4 + nc][anchors] of ONE frame — a YOLO-class graph's own decoded output (Ultralytics export: box centre / size in input pixels, class
 * probabilities) -> survivors of the same threshold / ordering / class-aware NMS as orc_yolo_post; returns count */
int orc_yolo_post_out(const float* out, int nc, int n_anchors, float conf, float iou, float* boxes5, int32_t* cls, int32_t* anchors, int cap) {
    std::vector<tk_yolo_cand_t> cand;
    for (int a = 0; a < n_anchors; ++a) {
        tk_yolo_cand_t c;
        const float cx = out[a], cy = out[(size_t)n_anchors + a], w = out[(size_t)2 * n_anchors + a], h = out[(size_t)3 * n_anchors + a];
        c.x1 = cx - 0.5f * w; c.y1 = cy - 0.5f * h; c.x2 = cx + 0.5f * w; c.y2 = cy + 0.5f * h;
        c.score = -1.0f; c.cls = 0;
        for (int k = 0; k < nc; ++k) {
            const float p = out[(size_t)(4 + k) * n_anchors + a];
            if (p > c.score) { c.score = p; c.cls = k; }
        }
        c.anchor = a;
        if (c.score > conf) cand.push_back(c);
    }
    std::stable_sort(cand.begin(), cand.end(), [](const tk_yolo_cand_t& p, const tk_yolo_cand_t& q) {
        return p.score > q.score || (p.score == q.score && p.anchor < q.anchor);
    });
    if ((int)cand.size() > TK_YOLO_MAX_CAND) cand.resize(TK_YOLO_MAX_CAND);
    std::vector<tk_yolo_cand_t> kept;
    for (const auto& c : cand) {
        bool drop = false;
        for (const auto& k : kept)
            if (k.cls == c.cls && tk_yolo_iou(&k, &c) > iou) { drop = true; break; }
        if (!drop) {
            kept.push_back(c);
            if ((int)kept.size() >= TK_OBJECT_DETECTOR_MAX_DETECTIONS) break;
        }
    }
    for (int i = 0; i < (int)kept.size() && i < cap; ++i) {
        boxes5[5 * i] = kept[i].x1; boxes5[5 * i + 1] = kept[i].y1; boxes5[5 * i + 2] = kept[i].x2; boxes5[5 * i + 3] = kept[i].y2;
        boxes5[5 * i + 4] = kept[i].score;
        cls[i] = kept[i].cls;
        anchors[i] = kept[i].anchor;
    }
    return (int)kept.size();
}

/* plain fp32 GEMM oracle of tk_gemm_f32: C = act(alpha * sum_k A[m][k] B[n][k] + bias) + residual, k-ordered fma chain */
void orc_gemm(const float* A, const float* B, float* C, const float* bias, const float* residual, int M, int N, int K, int lda, int ldb, int ldc,
              int ldr, int b_kn, int act, float alpha) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) acc = tk_fmaf(A[(size_t)m * lda + k], b_kn ? B[(size_t)k * ldb + n] : B[(size_t)n * ldb + k], acc);
            float v = acc;
            if (alpha != 1.0f) v = v * alpha;
            v = v + (bias ? bias[n] : 0.0f);
            if (act == 1) v = tk_siluf(v);
            else if (act == 2) v = tk_geluf(v);
            else if (act == 3) v = tk_sigmoidf(v);
            if (residual) v = v + residual[(size_t)m * ldr + n];
            C[(size_t)m * ldc + n] = v;
        }
}

} /* extern "C" */
