/* tk_fusion.cpp — see tk_fusion.h */
#include "tk_fusion.h"

#include <math.h>

#include <algorithm>

static uint32_t sat_u32(float v) { /* Rust's `as u32`: saturating, NaN -> 0 */
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)v;
}

float TkFusion::raw_distance(const TkBox& b, const float* depth, uint32_t dw, uint32_t dh, uint32_t frame_w, uint32_t frame_h) {
    if (!depth || dw == 0 || dh == 0) return -1.0f;
    const float nx0 = (float)b.x / (float)frame_w, ny0 = (float)b.y / (float)frame_h;
    const float nx1 = (float)(b.x + b.w) / (float)frame_w, ny1 = (float)(b.y + b.h) / (float)frame_h;
    const uint32_t x0 = sat_u32(roundf(nx0 * (float)(dw - 1))), y0 = sat_u32(roundf(ny0 * (float)(dh - 1)));
    const uint32_t x1 = sat_u32(roundf(nx1 * (float)(dw - 1))), y1 = sat_u32(roundf(ny1 * (float)(dh - 1)));
    if (x0 >= dw || y0 >= dh || x1 >= dw || y1 >= dh || x0 >= x1 || y0 >= y1) return -1.0f;
    std::vector<float> v;
    for (uint32_t y = y0; y <= y1; ++y)
        for (uint32_t x = x0; x <= x1; ++x) {
            const float d = depth[(size_t)y * dw + x];
            if (d > 0.1f && d < 100.0f) v.push_back(d);
        }
    if (v.size() < 10) return -1.0f;
    std::sort(v.begin(), v.end());
    const float q1 = v[v.size() / 4], q3 = v[v.size() * 3 / 4], iqr = q3 - q1;
    const float lo = q1 - 1.5f * iqr, hi = q3 + 1.5f * iqr;
    float sum = 0.0f;
    size_t cnt = 0;
    for (float d : v)
        if (d >= lo && d <= hi) { sum = sum + d; ++cnt; }
    return cnt ? sum / (float)cnt : -1.0f;
}

float TkFusion::iou(const TkBox& a, const TkBox& b) {
    const int xl = std::max(a.x, b.x), yt = std::max(a.y, b.y), xr = std::min(a.x + a.w, b.x + b.w), yb = std::min(a.y + a.h, b.y + b.h);
    if (xr < xl || yb < yt) return 0.0f;
    const float inter = (float)(xr - xl) * (float)(yb - yt);
    const float uni = (float)(a.w * a.h) + (float)(b.w * b.h) - inter;
    return uni > 0.0f ? inter / uni : 0.0f;
}

void TkFusion::fuse(const TkBox* boxes, const uint32_t* class_ids, size_t n, const float* depth, uint32_t dw, uint32_t dh, uint32_t frame_w, uint32_t frame_h,
                    float fx, float fy, std::vector<TkFused>* out) {
    out->assign(n, TkFused());
    for (auto& t : tr_) t.matched = false;
    std::vector<uint64_t> served(n, 0);
    for (size_t i = 0; i < n; ++i) {
        const float raw = raw_distance(boxes[i], depth, dw, dh, frame_w, frame_h);
        if (raw < 0.0f) continue;
        int best = -1;
        float best_iou = 0.0f;
        for (size_t t = 0; t < tr_.size(); ++t) {
            const float v = iou(boxes[i], tr_[t].last);
            if (v > 0.4f && v > best_iou) { best = (int)t; best_iou = v; }
        }
        if (best >= 0) {
            Tracker& t = tr_[(size_t)best];
            t.p = t.p + 0.1f;                       /* predict: x stays, P = F P F' + Q */
            const float k = t.p / (t.p + 0.5f);     /* update: K = P H' / (H P H' + R) */
            t.x = t.x + k * (raw - t.x);
            t.p = (1.0f - k) * t.p;
            t.last = boxes[i];
            t.unseen = 0;
            t.matched = true;
            served[i] = t.id;
        } else {
            Tracker t{class_ids ? class_ids[i] : 0u, boxes[i], raw, 1.0f, 0u, true, next_id_++};
            tr_.push_back(t);
            served[i] = t.id;
        }
    }
    for (size_t i = 0; i < n; ++i) {
        if (!served[i]) continue;
        for (size_t t = 0; t < tr_.size(); ++t)
            if (tr_[t].id == served[i]) {
                TkFused& f = (*out)[i];
                f.valid = true;
                f.tracker_id = tr_[t].id;
                f.distance_m = tr_[t].x; /* the smoothed distance after every detection of this frame has been absorbed */
                if (f.distance_m > 0.0f) { f.width_m = (float)tr_[t].last.w * f.distance_m / fx; f.height_m = (float)tr_[t].last.h * f.distance_m / fy; }
                else { f.width_m = f.height_m = -1.0f; }
            }
    }
    std::vector<Tracker> keep;
    for (auto& t : tr_) {
        if (!t.matched) { t.unseen += 1; if (t.unseen > 5) continue; }
        keep.push_back(t);
    }
    tr_.swap(keep);
}
