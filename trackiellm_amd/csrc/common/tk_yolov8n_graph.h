/*
 * tk_yolov8n_graph.h — topology of the YOLOv8n detector, written once against an abstract
 * "Ops" backend so the HIP engine and the CPU oracle walk the SAME graph (the independent
 * torch implementation in tests/golden/make_vision_golden.py pins the topology itself).
 *
 * The reference never sees this graph: it hands an .onnx file to ONNX Runtime
 * (src/vision/tk_object_detector.c:93-152, run at :261-301).  north_star names YOLOv8n; the
 * reference's file names say yolov5nu — both export the same
 * [1, 84, 8400] head, which is what is built here: Ultralytics YOLOv8n, width 0.25, depth 0.33,
 * BatchNorm folded into conv bias, SiLU, C2f / SPPF blocks, decoupled DFL head (reg_max 16).
 *
 * Activations are NHWC fp32; a TkT view may be a channel slice of a wider buffer (ld > C), which
 * makes every concat / chunk in the graph free.
 */
#ifndef TK_YOLOV8N_GRAPH_H
#define TK_YOLOV8N_GRAPH_H

#include <stdint.h>

#include <vector>

struct TkT {
    float* p = nullptr;
    int B = 0, H = 0, W = 0, C = 0;
    int ld = 0; /* floats between consecutive pixels */
    TkT slice(int c0, int c) const { TkT t = *this; t.p = p + c0; t.C = c; return t; }
};

struct TkConvSpec { int cin, cout, k, s, act; };

enum { TK_YOLO_REG_MAX = 16, TK_YOLO_NC = 80, TK_YOLO_NO = 4 * TK_YOLO_REG_MAX + TK_YOLO_NC };

/*
 * Ops concept:
 *   TkT  alloc(int B, int H, int W, int C);                       // scratch for this forward
 *   void conv(const TkT& x, int idx, const TkT& y, const TkT* residual);   // layer idx of specs()
 *   void maxpool5(const TkT& x, const TkT& y);
 *   void upsample2x(const TkT& x, const TkT& y);
 *   void copy(const TkT& x, const TkT& y);
 */
template <class Ops>
class TkYoloV8n {
public:
    /* the 63 conv layers in call order */
    static std::vector<TkConvSpec> specs(int nc = TK_YOLO_NC) {
        std::vector<TkConvSpec> v;
        Builder b{&v};
        walk(b, nc);
        return v;
    }

    /* x: [B, H, W, 3] (H, W multiples of 32); out[i]: [B, H/s, W/s, 64 + nc] raw head maps for s = 8, 16, 32 */
    static void forward(Ops& ops, const TkT& x, TkT out[3], int nc = TK_YOLO_NC) {
        Runner r{&ops, x, out, 0};
        walk(r, nc);
    }

private:
    struct Builder {
        std::vector<TkConvSpec>* v;
        TkT alloc(int B, int H, int W, int C) { TkT t; t.B = B; t.H = H; t.W = W; t.C = C; t.ld = C; return t; }
        void conv(const TkT& x, const TkT& y, int k, int s, int act, const TkT*) { v->push_back(TkConvSpec{x.C, y.C, k, s, act}); }
        void maxpool5(const TkT&, const TkT&) {}
        void upsample2x(const TkT&, const TkT&) {}
        void copy(const TkT&, const TkT&) {}
        TkT input() { return alloc(1, 64, 64, 3); }
        void output(int, const TkT&) {}
    };
    struct Runner {
        Ops* ops;
        TkT in;
        TkT* out;
        int idx;
        TkT alloc(int B, int H, int W, int C) { return ops->alloc(B, H, W, C); }
        void conv(const TkT& x, const TkT& y, int, int, int, const TkT* res) { ops->conv(x, idx++, y, res); }
        void maxpool5(const TkT& x, const TkT& y) { ops->maxpool5(x, y); }
        void upsample2x(const TkT& x, const TkT& y) { ops->upsample2x(x, y); }
        void copy(const TkT& x, const TkT& y) { ops->copy(x, y); }
        TkT input() { return in; }
        void output(int i, const TkT& t) { out[i] = t; }
    };

    template <class G>
    static TkT conv(G& g, const TkT& x, int cout, int k, int s, int act = 1) {
        TkT y = g.alloc(x.B, (x.H + 2 * (k / 2) - k) / s + 1, (x.W + 2 * (k / 2) - k) / s + 1, cout);
        g.conv(x, y, k, s, act, nullptr);
        return y;
    }

    template <class G>
    static TkT c2f(G& g, const TkT& x, int c2, int n, bool shortcut) {
        const int c = c2 / 2;
        TkT cat = g.alloc(x.B, x.H, x.W, (2 + n) * c);
        g.conv(x, cat.slice(0, 2 * c), 1, 1, 1, nullptr);
        for (int i = 0; i < n; ++i) {
            TkT in = cat.slice((i + 1) * c, c);
            TkT t = g.alloc(x.B, x.H, x.W, c);
            g.conv(in, t, 3, 1, 1, nullptr);
            g.conv(t, cat.slice((i + 2) * c, c), 3, 1, 1, shortcut ? &in : nullptr);
        }
        TkT y = g.alloc(x.B, x.H, x.W, c2);
        g.conv(cat, y, 1, 1, 1, nullptr);
        return y;
    }

    template <class G>
    static TkT sppf(G& g, const TkT& x, int c2) {
        const int c = x.C / 2;
        TkT cat = g.alloc(x.B, x.H, x.W, 4 * c);
        g.conv(x, cat.slice(0, c), 1, 1, 1, nullptr);
        for (int i = 0; i < 3; ++i) g.maxpool5(cat.slice(i * c, c), cat.slice((i + 1) * c, c));
        TkT y = g.alloc(x.B, x.H, x.W, c2);
        g.conv(cat, y, 1, 1, 1, nullptr);
        return y;
    }

    template <class G>
    static TkT up_cat(G& g, const TkT& low, const TkT& skip) {
        TkT cat = g.alloc(low.B, 2 * low.H, 2 * low.W, low.C + skip.C);
        g.upsample2x(low, cat.slice(0, low.C));
        g.copy(skip, cat.slice(low.C, skip.C));
        return cat;
    }

    template <class G>
    static TkT down_cat(G& g, const TkT& hi, const TkT& skip) {
        TkT cat = g.alloc(skip.B, skip.H, skip.W, hi.C + skip.C);
        g.conv(hi, cat.slice(0, hi.C), 3, 2, 1, nullptr);
        g.copy(skip, cat.slice(hi.C, skip.C));
        return cat;
    }

    template <class G>
    static void walk(G& g, int nc) {
        TkT x = g.input();
        x = conv(g, x, 16, 3, 2);
        x = conv(g, x, 32, 3, 2);
        x = c2f(g, x, 32, 1, true);
        x = conv(g, x, 64, 3, 2);
        TkT f4 = c2f(g, x, 64, 2, true);
        x = conv(g, f4, 128, 3, 2);
        TkT f6 = c2f(g, x, 128, 2, true);
        x = conv(g, f6, 256, 3, 2);
        x = c2f(g, x, 256, 1, true);
        TkT f9 = sppf(g, x, 256);
        TkT h12 = c2f(g, up_cat(g, f9, f6), 128, 1, false);
        TkT h15 = c2f(g, up_cat(g, h12, f4), 64, 1, false);
        TkT h18 = c2f(g, down_cat(g, h15, h12), 128, 1, false);
        TkT h21 = c2f(g, down_cat(g, h18, f9), 256, 1, false);
        const TkT feats[3] = {h15, h18, h21};
        const int c2 = 64, c3 = nc > 64 ? nc : 64; /* max(16, ch0/4, 4*reg_max) and max(ch0, min(nc, 100)) for YOLOv8n */
        for (int i = 0; i < 3; ++i) {
            TkT o = g.alloc(feats[i].B, feats[i].H, feats[i].W, 4 * TK_YOLO_REG_MAX + nc);
            TkT b = conv(g, feats[i], c2, 3, 1);
            b = conv(g, b, c2, 3, 1);
            g.conv(b, o.slice(0, 4 * TK_YOLO_REG_MAX), 1, 1, 0, nullptr);
            TkT c = conv(g, feats[i], c3, 3, 1);
            c = conv(g, c, c3, 3, 1);
            g.conv(c, o.slice(4 * TK_YOLO_REG_MAX, nc), 1, 1, 0, nullptr);
            g.output(i, o);
        }
    }
};

#endif
