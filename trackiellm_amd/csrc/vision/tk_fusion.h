/*
 * tk_fusion.h — object / depth fusion: a distance, a metric width and height for every detected box (host code, a few boxes per frame).
 *
 * Restates src/vision/src/object_analysis.rs (the Rust half of fuse_object_depth, src/vision/tk_vision_pipeline.c:653-713):
 *   calculate_raw_distance (:227-279): the box is scaled into the depth map (normalised corners * (dim - 1), rounded), depths in
 *   (0.1, 100) are collected row by row; with >= 10 of them the inter-quartile filter [q1 - 1.5 iqr, q3 + 1.5 iqr] is applied
 *   (q1 = sorted[len / 4], q3 = sorted[len * 3 / 4]) and the mean of the survivors is the raw distance; otherwise -1.
 *   fuse_object_and_depth_data (:134-223): a detection matches the tracker with the highest IoU > 0.4 (integer boxes); a matched tracker
 *   runs one scalar Kalman step (F = H = 1, Q = 0.1, R = 0.5, P0 = 1; predict, then update with the raw distance), an unmatched
 *   detection starts a tracker at its raw distance; trackers unseen for more than 5 frames are dropped; width = w * distance / fx,
 *   height = h * distance / fy.
 * Deviations, both deterministic where the reference is not: trackers are kept in creation order (the reference iterates a HashMap keyed
 * by random UUIDs), and results are reported per detection in detection order (the reference returns one entry per matched tracker in
 * HashMap order and the C caller copies them positionally).  include/tk/ABI_NOTES.md records this.
 */
#ifndef TK_FUSION_H
#define TK_FUSION_H

#include <stdint.h>

#include <vector>

struct TkBox { int x, y, w, h; };

struct TkFused {
    float distance_m = 0.0f, width_m = 0.0f, height_m = 0.0f; /* 0: no valid depth under the box */
    bool valid = false;
    uint64_t tracker_id = 0; /* the tracker that served this detection */
};

class TkFusion {
public:
    struct Tracker { uint32_t class_id; TkBox last; float x, p; uint32_t unseen; bool matched; uint64_t id; };
    static float raw_distance(const TkBox& b, const float* depth, uint32_t dw, uint32_t dh, uint32_t frame_w, uint32_t frame_h);
    static float iou(const TkBox& a, const TkBox& b);
    void fuse(const TkBox* boxes, const uint32_t* class_ids, size_t n, const float* depth, uint32_t dw, uint32_t dh, uint32_t frame_w, uint32_t frame_h, float fx,
              float fy, std::vector<TkFused>* out);
    const std::vector<Tracker>& trackers() const { return tr_; }
    void clear() { tr_.clear(); }

private:
    std::vector<Tracker> tr_;
    uint64_t next_id_ = 1;
};

#endif
