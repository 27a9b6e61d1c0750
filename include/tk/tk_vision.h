/*
 * tk_vision.h — detector stream of the tk_* C-ABI.
 *   tk_object_detector_*                         src/vision/tk_object_detector.h:42-57 (config), :66-71 (result), :99-166 (API)
 *   tk_preprocessor_resize_and_normalize_to_chw  src/vision/tk_image_preprocessor.h:49-56
 *   tk_classify_dominant_color / _door_state    src/vision/tk_attribute_classifier.h (per-box attributes, tk_vision_pipeline.c:462-485)
 *   tk_vision_pipeline_* / tk_vision_result_*    src/vision/tk_vision_pipeline.h:118-335: object detection, depth estimation and the
 *                                                object / depth fusion (tk_depth.h); OCR, navigation cues and the scene graph are out of scope
 * model_path forms: an Ultralytics YOLOv8n .onnx, a TKYOLO1 weight container or "synthetic://yolov8n?seed=5&cls_bias=-4"
 * (the network topology is fixed: YOLOv8n, nc = class_count).
 * Results: score-descending, at most 500, class-aware NMS applied, bbox in ORIGINAL frame pixels.
 * `label` is borrowed from config.class_labels (never freed by the library; the reference's
 * tk_vision_result_destroy frees it wrongly, src/vision/tk_vision_pipeline.c:297-299 — not replicated).
 */
#ifndef TK_MI355X_VISION_H
#define TK_MI355X_VISION_H

#include "tk_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tk_object_detector_s tk_object_detector_t;

typedef struct {
    tk_vision_backend_e backend; /* anything but CPU selects the GPU; CPU is refused: no fallback exists */
    int gpu_device_id;
    tk_path_t* model_path;
    uint32_t input_width;
    uint32_t input_height;
    const char** class_labels;
    size_t class_count;
    float confidence_threshold;
    float iou_threshold;
} tk_object_detector_config_t;

typedef struct {
    uint32_t class_id;
    const char* label;
    float confidence;
    tk_rect_t bbox;
} tk_detection_result_t;

TK_API TK_NODISCARD tk_error_code_t tk_object_detector_create(tk_object_detector_t** out_detector, const tk_object_detector_config_t* config);
TK_API void tk_object_detector_destroy(tk_object_detector_t** detector);
TK_API TK_NODISCARD tk_error_code_t tk_object_detector_detect(tk_object_detector_t* detector, const tk_video_frame_t* video_frame,
                                                              tk_detection_result_t** out_results, size_t* out_result_count);
TK_API void tk_object_detector_free_results(tk_detection_result_t** results);
TK_API void tk_object_detector_update_thresholds(tk_object_detector_t* detector, float confidence_threshold, float iou_threshold);

/* host buffers in, host buffer out; the arithmetic runs on the GPU and is bit-identical to the
 * reference's scalar CPU loop (frame->stride is honoured; the reference assumes width*3) */
TK_API TK_NODISCARD tk_error_code_t tk_preprocessor_resize_and_normalize_to_chw(const tk_video_frame_t* frame, float* out_tensor,
                                                                                uint32_t target_width, uint32_t target_height,
                                                                                const float mean[3], const float std_dev[3]);

/* ---- per-box attributes: host frame in, strdup'ed name out (caller frees), computed on the GPU, bit-identical to the reference's CPU
 * loops.  Like the reference the frame is read as tightly packed RGB8 (width * 3 bytes per row). ---- */
TK_API TK_NODISCARD tk_error_code_t tk_classify_dominant_color(const tk_video_frame_t* frame, const tk_rect_t* bbox, char** out_color_name);
TK_API TK_NODISCARD tk_error_code_t tk_classify_door_state(const tk_video_frame_t* frame, const tk_rect_t* bbox, char** out_state_name);

/* ---- vision pipeline: object detection, depth estimation, object / depth fusion (OCR / navigation cues / scene graph are out of scope:
 * their flags are accepted and never set in valid_analyses_mask) ---- */
typedef struct tk_vision_pipeline_s tk_vision_pipeline_t;
typedef struct tk_vision_result_s tk_vision_result_t;
typedef uint32_t tk_vision_analysis_flags_t;
enum {
    TK_VISION_ANALYZE_NONE = 0, TK_VISION_ANALYZE_OBJECT_DETECTION = 1 << 0, TK_VISION_ANALYZE_DEPTH_ESTIMATION = 1 << 1,
    TK_VISION_ANALYZE_OCR = 1 << 2, TK_VISION_ANALYZE_FUSION_DISTANCE = 1 << 3, TK_VISION_ANALYZE_NAVIGATION_CUES = 1 << 4,
    TK_VISION_ANALYZE_SCENE_GRAPH = 1 << 5
};
typedef uint32_t tk_vision_valid_result_flags_t;
enum {
    TK_VISION_RESULT_NONE = 0, TK_VISION_RESULT_OBJECT_DETECTION = 1 << 0, TK_VISION_RESULT_DEPTH_ESTIMATION = 1 << 1,
    TK_VISION_RESULT_OCR = 1 << 2, TK_VISION_RESULT_FUSION_DISTANCE = 1 << 3, TK_VISION_RESULT_NAVIGATION_CUES = 1 << 4
};
typedef struct {
    tk_vision_backend_e backend;
    int gpu_device_id;
    tk_path_t* object_detection_model_path;
    tk_path_t* depth_estimation_model_path; /* NULL: no depth analysis; run at 256 x 256 as the reference does (tk_vision_pipeline.c:388-394) */
    tk_path_t* tesseract_data_path;         /* ignored */
    float object_confidence_threshold;
    uint32_t max_detected_objects;
    float focal_length_x;
    float focal_length_y;
} tk_vision_pipeline_config_t;
typedef struct {
    float object_confidence_threshold;
    float iou_threshold;
    bool enable_object_detection;
    bool enable_depth_estimation;
} tk_vision_runtime_config_t;
typedef struct {
    uint32_t class_id;
    const char* label;      /* owned by the object here (a copy): tk_vision_result_destroy frees it, as the reference's does */
    float confidence;
    tk_rect_t bbox;
    float distance_meters;  /* set by TK_VISION_ANALYZE_FUSION_DISTANCE (Kalman-smoothed); 0 when the box has no valid depth */
    float width_meters;
    float height_meters;
    bool is_partially_occluded;
    char* recognized_text;  /* NULL: OCR is out of scope */
    char* attributes;       /* "color:<name>" (+ ",state:open|closed" for labels containing "door"); owned by the object */
} tk_vision_object_t;
typedef struct { char* text; float confidence; tk_rect_t bbox; } tk_vision_text_block_t;
typedef struct { uint32_t width; uint32_t height; float* data; } tk_vision_depth_map_t;
struct tk_vision_result_s {
    uint64_t source_frame_timestamp_ns;
    tk_vision_valid_result_flags_t valid_analyses_mask;
    size_t object_count;
    tk_vision_object_t* objects;
    size_t text_block_count;
    tk_vision_text_block_t* text_blocks; /* always NULL */
    tk_vision_depth_map_t* depth_map;    /* TK_VISION_ANALYZE_DEPTH_ESTIMATION: metres, owned by the result */
    char* serialized_scene_graph;        /* always NULL */
};
TK_API TK_NODISCARD tk_error_code_t tk_vision_pipeline_create(tk_vision_pipeline_t** out_pipeline, const tk_vision_pipeline_config_t* config);
TK_API void tk_vision_pipeline_destroy(tk_vision_pipeline_t** pipeline);
TK_API TK_NODISCARD tk_error_code_t tk_vision_pipeline_update_config(tk_vision_pipeline_t* pipeline, const tk_vision_runtime_config_t* config);
TK_API TK_NODISCARD tk_error_code_t tk_vision_pipeline_process_frame(tk_vision_pipeline_t* pipeline, const tk_video_frame_t* video_frame,
                                                                     tk_vision_analysis_flags_t analysis_flags, const tk_rect_t* ocr_roi,
                                                                     uint64_t timestamp_ns, tk_vision_result_t** out_result);
TK_API void tk_vision_result_destroy(tk_vision_result_t** result);

/* ---- extensions (no reference counterpart) ---- */
/* B frames of identical geometry in one pass (one per concurrent cortex cycle); results[i] / counts[i] per frame; max_batch in [1, 256] */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_detector_set_max_batch(tk_object_detector_t* detector, int max_batch);
/* the per-model-file registry behind tk_object_detector_create: handles opened on the same file / device / geometry share the weights and one
 * batched engine whose scheduler coalesces their one-frame calls (tk_object_detector_detect, tk_vision_pipeline_process_frame).  Counters of
 * this handle's shared engine: live handles, batched jobs run, frames they carried, the widest job.  Any pointer may be NULL. */
/* Opt-in fast contraction: the detector's convolutions run on the f16 matrix pipe with every operand split into two f16
 * halves (~22 significant bits, fp32 accumulation) instead of the exact fp32 chain — results within ~1e-6 of the chain's scale, not its bits.
 * Off by default; the exact path stays the parity path and the checker (tests/test_vision_gpu.py::test_detector_fast_contraction_gate).
 * Handles of one model file with different settings still share the engine; their frames ride separate jobs.  A fast handle's result for a
 * frame does not depend on the frames that share its job (the kernel that evaluates a layer is chosen by the layer, never by the batch). */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_detector_set_fast_contraction(tk_object_detector_t* detector, int on);
TK_API void tk_mi355x_detector_share_stats(const tk_object_detector_t* detector, uint64_t* handles, uint64_t* batches, uint64_t* frames, uint64_t* widest);
/* ONNX detector file (tk_object_detector_config_t.model_path, src/vision/tk_object_detector.c:93-152): parse the Conv initialisers
 * (no ONNX Runtime, no GPU) and check them against the YOLOv8n graph this path runs; n_convs / n_params optional */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_onnx_probe(const char* path, int32_t* n_convs, int64_t* n_params);
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_detector_detect_batch(tk_object_detector_t* detector, int n_frames, const tk_video_frame_t* frames,
                                                                    tk_detection_result_t** out_results, size_t* out_counts);
/* A detector .onnx that is not the 63-convolution YOLOv8n topology (the reference names yolov5nu.onnx, src/cortex/tk_cortex_main.h:71) runs its own
 * graph on the library's ONNX executor and its [1, 4 + nc, anchors] output is decoded and NMS-ed like the YOLOv8n head maps: 1 for such a handle */
TK_API int tk_mi355x_detector_is_graph(const tk_object_detector_t* detector);
/* test hook of that path: the graph on a pre-processed planar tensor [B][3][H][W]; out [B][4 + nc][anchors] = the file's own output */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_detector_forward_graph(tk_object_detector_t* detector, int batch, const float* nchw, float* out, size_t out_floats);
/* test hook: run the network on a pre-processed NHWC fp32 tensor [B][H][W][3]; raw head maps [B][anchors][64+nc] */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_detector_forward_raw(tk_object_detector_t* detector, int batch, const float* nhwc, float* raw_out,
                                                                   size_t raw_floats);
/* detections of the last forward_raw / detect in input-tensor coordinates: [x1,y1,x2,y2,score] + class + anchor */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_detector_last_boxes(tk_object_detector_t* detector, int frame, float* boxes5, int32_t* cls,
                                                                  int32_t* anchors, int cap, int* count);
TK_API int tk_mi355x_detector_anchor_count(const tk_object_detector_t* detector);

#ifdef __cplusplus
}
#endif
#endif
