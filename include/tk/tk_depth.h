/*
 * tk_depth.h — monocular depth estimation and object / depth fusion of the vision stream.
 *   tk_depth_estimator_*            src/vision/tk_depth_midas.h:42-50 (config), :75-131 (API); implementation src/vision/tk_depth_midas.c
 *   tk_vision_rust_fuse_data / _free_fused_result
 *                                   src/vision/src/lib.rs:173-253 (C-ABI exported by the reference's Rust crate, called from
 *                                   fuse_object_depth, src/vision/tk_vision_pipeline.c:653-713); logic src/vision/src/object_analysis.rs
 * model_path: an ONNX file of the convolutional MiDaS class (MiDaS v2.1 small / large: Conv with groups, Relu / Clip, Add, Concat,
 * Resize, pooling, BatchNormalization — the op list of csrc/nn/tk_onnx_exec.h), input "[1, 3, input_height, input_width]" float,
 * first output [1, H, W] or [1, 1, H, W] with H x W = input_height x input_width.  A DPT / Swin transformer export uses ops outside that
 * list and is refused at create time with the op's name (TK_ERROR_MODEL_LOAD_FAILED) — the reference's header names DPT-SwinV2-Tiny, its
 * code accepts any ONNX file.  The depth map is input_width x input_height floats in metres (0.1 .. 10, nearer = larger raw response),
 * malloc'ed for the caller: free with tk_depth_estimator_free_map (or tk_vision_result_destroy when it sits in a pipeline result).
 */
#ifndef TK_MI355X_DEPTH_H
#define TK_MI355X_DEPTH_H

#include "tk_vision.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tk_depth_estimator_s tk_depth_estimator_t;

typedef struct {
    tk_vision_backend_e backend; /* anything but CPU selects the GPU; CPU is refused: no fallback exists */
    int gpu_device_id;
    tk_path_t* model_path;
    uint32_t input_width;
    uint32_t input_height;
} tk_depth_estimator_config_t;

TK_API TK_NODISCARD tk_error_code_t tk_depth_estimator_create(tk_depth_estimator_t** out_estimator, const tk_depth_estimator_config_t* config);
TK_API void tk_depth_estimator_destroy(tk_depth_estimator_t** estimator);
TK_API TK_NODISCARD tk_error_code_t tk_depth_estimator_estimate(tk_depth_estimator_t* estimator, const tk_video_frame_t* video_frame,
                                                                tk_vision_depth_map_t** out_depth_map);
TK_API void tk_depth_estimator_free_map(tk_vision_depth_map_t** depth_map);

/* ---- fusion: the C-ABI the reference's Rust crate exports (EnrichedObject / CFusedResult are #[repr(C)] there) ---- */
typedef struct {
    uint32_t class_id;
    float confidence;      /* 1.0: the reference reports the tracker, not the detection (object_analysis.rs:201) */
    tk_rect_t bbox;
    float distance_meters;
    float width_meters;
    float height_meters;
    bool is_partially_occluded; /* always false (object_analysis.rs:206) */
} tk_enriched_object_t;
typedef struct {
    const tk_enriched_object_t* objects;
    size_t count;
} tk_fused_result_t;
/* process-wide trackers, as the reference's lazy_static TRACKERS; one entry per detection that has valid depth under its box, in
 * detection order (the reference's order is that of a HashMap keyed by random UUIDs) */
TK_API tk_fused_result_t* tk_vision_rust_fuse_data(const tk_detection_result_t* detections, size_t detection_count, const tk_vision_depth_map_t* depth_map,
                                                   uint32_t frame_width, uint32_t frame_height, float focal_length_x, float focal_length_y);
TK_API void tk_vision_rust_free_fused_result(tk_fused_result_t* result);

/* ---- extensions (no reference counterpart) ---- */
/* ONNX depth model file: parse it (no GPU) and check every op against the executor's list; n_nodes optional */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_depth_onnx_probe(const char* path, int32_t* n_nodes);
/* test hooks: the network alone on a pre-processed planar tensor [3][input_height][input_width] -> raw output [input_height][input_width];
 * the raw output of the last tk_depth_estimator_estimate call; forget the process-wide fusion trackers */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_depth_forward_raw(tk_depth_estimator_t* estimator, const float* chw, float* raw_out, size_t raw_floats);
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_depth_last_raw(tk_depth_estimator_t* estimator, float* raw_out, size_t raw_floats);
TK_API void tk_mi355x_fusion_reset(void);
/* raw (unsmoothed) distance under one box: calculate_raw_distance, object_analysis.rs:227-279; -1 when fewer than 10 valid depths */
TK_API float tk_mi355x_fusion_raw_distance(const tk_rect_t* bbox, const tk_vision_depth_map_t* depth_map, uint32_t frame_width, uint32_t frame_height);

#ifdef __cplusplus
}
#endif

#endif
