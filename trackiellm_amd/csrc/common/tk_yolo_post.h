/*
 * tk_yolo_post.h — detector post-processing arithmetic shared by the HIP kernels and the oracle.
 *
 * The reference's postprocess_detections is a stub (src/vision/tk_object_detector.c:303-368: objectness
 * threshold only, "NMS logic would go here", calculate_iou returns 0, coordinates never rescaled —
 * SURVEY.md §0 F3), while its header promises threshold + NMS + conversion to the original frame
 * (src/vision/tk_object_detector.h:118-143).  This restates what the header promises for a
 * YOLOv8 head ([64 DFL logits | nc class logits] per anchor):
 *   - DFL expectation over 16 bins per side, anchor centre (x+0.5, y+0.5), xyxy = (a -/+ d) * stride
 *   - score = max_c sigmoid(cls_c) (first class on ties); candidate iff score > confidence_threshold
 *     (the reference's comparison is a strict '>', tk_object_detector.c:338)
 *   - candidates ordered by (score desc, anchor index asc), at most TK_YOLO_MAX_CAND enter NMS
 *   - greedy class-aware NMS: a box is dropped iff IoU with an already kept box of the SAME class > iou_threshold
 *   - at most TK_OBJECT_DETECTOR_MAX_DETECTIONS (500, tk_object_detector.c:39) results, score-descending
 *   - bbox is mapped to the original frame by the stretch factors (the reference pre-processor does not
 *     letterbox) and truncated to int like the reference's (int) casts (:352-355)
 */
#ifndef TK_YOLO_POST_H
#define TK_YOLO_POST_H

#include "tk_exact_math.h"

#define TK_YOLO_MAX_CAND 2048
#define TK_OBJECT_DETECTOR_MAX_DETECTIONS 500

typedef struct {
    float x1, y1, x2, y2; /* input-tensor pixels */
    float score;
    int32_t cls;
    int32_t anchor;
} tk_yolo_cand_t;

TK_HD float tk_yolo_dfl(const float* l) {
    float m = l[0];
    for (int i = 1; i < 16; ++i) m = tk_fmaxf(m, l[i]);
    float s = 0.0f, e = 0.0f;
    for (int i = 0; i < 16; ++i) {
        const float p = tk_expf(l[i] - m);
        s = s + p;
        e = tk_fmaf((float)i, p, e);
    }
    return tk_divf(e, s);
}

/* o: 64 + nc raw head values of one anchor (may be strided by `ld`) */
TK_HD void tk_yolo_decode_anchor(const float* o, int nc, float ax, float ay, float stride, tk_yolo_cand_t* c) {
    float d[4];
    for (int k = 0; k < 4; ++k) d[k] = tk_yolo_dfl(o + 16 * k);
    c->x1 = (ax - d[0]) * stride;
    c->y1 = (ay - d[1]) * stride;
    c->x2 = (ax + d[2]) * stride;
    c->y2 = (ay + d[3]) * stride;
    float best = -1.0f;
    int bi = 0;
    for (int k = 0; k < nc; ++k) {
        const float s = tk_sigmoidf(o[64 + k]);
        if (s > best) { best = s; bi = k; }
    }
    c->score = best;
    c->cls = bi;
}

/* o: column `anchor` of a [4 + nc][n_anchors] decoded output (row pitch ld): centre x, centre y, width, height, then nc class probabilities */
TK_HD void tk_yolo_decode_out_anchor(const float* o, int ld, int nc, tk_yolo_cand_t* c) {
    const float cx = o[0], cy = o[(size_t)ld], w = o[(size_t)2 * ld], h = o[(size_t)3 * ld];
    c->x1 = cx - 0.5f * w;
    c->y1 = cy - 0.5f * h;
    c->x2 = cx + 0.5f * w;
    c->y2 = cy + 0.5f * h;
    float best = -1.0f;
    int bi = 0;
    for (int k = 0; k < nc; ++k) {
        const float s = o[(size_t)(4 + k) * ld];
        if (s > best) { best = s; bi = k; }
    }
    c->score = best;
    c->cls = bi;
}

TK_HD float tk_yolo_iou(const tk_yolo_cand_t* a, const tk_yolo_cand_t* b) {
    const float iw = tk_fminf(a->x2, b->x2) - tk_fmaxf(a->x1, b->x1);
    const float ih = tk_fminf(a->y2, b->y2) - tk_fmaxf(a->y1, b->y1);
    if (!(iw > 0.0f) || !(ih > 0.0f)) return 0.0f;
    const float inter = iw * ih;
    const float ua = (a->x2 - a->x1) * (a->y2 - a->y1);
    const float ub = (b->x2 - b->x1) * (b->y2 - b->y1);
    const float uni = (ua + ub) - inter;
    return uni > 0.0f ? tk_divf(inter, uni) : 0.0f;
}

/* synthetic detector weights (no .onnx offline — SURVEY.md §8d): He-scaled normals, small biases, the
 * last class conv of each scale gets `cls_bias` so a random network still emits a handful of boxes */
TK_HD float tk_yolo_synth_w(uint64_t seed, int layer, int64_t index, int fan_in) {
    const float sd = tk_sqrtf(tk_divf(2.0f, (float)fan_in));
    return sd * tk_synth_normal(seed, (uint64_t)(1000 + layer), (uint64_t)index);
}
TK_HD float tk_yolo_synth_b(uint64_t seed, int layer, int64_t index, int is_cls_out, float cls_bias) {
    const float b = 0.05f * tk_synth_normal(seed, (uint64_t)(5000 + layer), (uint64_t)index);
    return is_cls_out ? b + cls_bias : b;
}

#endif
