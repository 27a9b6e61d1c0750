/*
 * tk_abi_depth.cpp — tk_depth_estimator_* (src/vision/tk_depth_midas.h:75-131) on the HIP depth engine, and the fusion C-ABI of the
 * reference's Rust vision crate (src/vision/src/lib.rs:173-253).
 */
#include <stdlib.h>
#include <string.h>

#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../nn/tk_onnx_graph.h"
#include "../vision/tk_depth_engine.h"
#include "../vision/tk_fusion.h"
#include "tk/tk_depth.h"
#include "tk/tk_error_handling.h"

struct tk_depth_estimator_s {
    TkDepthEngine eng;
    std::vector<float> last_raw;
};

static tk_error_code_t dfail(tk_error_code_t code, const std::string& why) {
    tk_error_set_detail("%s", why.c_str());
    return code;
}

static std::mutex g_fusion_mu;
static TkFusion g_fusion;

extern "C" {

tk_error_code_t tk_depth_estimator_create(tk_depth_estimator_t** out_estimator, const tk_depth_estimator_config_t* config) {
    if (!out_estimator || !config || !config->model_path || !config->model_path->path_str) return TK_ERROR_INVALID_ARGUMENT;
    if (config->input_width == 0 || config->input_height == 0) return TK_ERROR_INVALID_ARGUMENT; /* tk_depth_midas.c:99-101 */
    if (config->input_width > 2048 || config->input_height > 2048) return dfail(TK_ERROR_INVALID_ARGUMENT, "input dimensions above 2048");
    if (config->backend == TK_VISION_BACKEND_CPU) return dfail(TK_ERROR_BACKEND_NOT_SUPPORTED, "this library is the MI355X path: there is no CPU backend");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return dfail(TK_ERROR_GPU_DEVICE_NOT_FOUND, "no HIP device visible");
    const int dev = config->gpu_device_id < 0 ? 0 : config->gpu_device_id;
    if (dev >= ndev) return dfail(TK_ERROR_INVALID_ARGUMENT, "gpu_device_id out of range");
    std::unique_ptr<tk_depth_estimator_s> e(new tk_depth_estimator_s());
    try {
        if (!e->eng.load(config->model_path->path_str, dev, config->input_width, config->input_height)) return dfail(TK_ERROR_MODEL_LOAD_FAILED, e->eng.error);
    } catch (const std::exception& ex) { return dfail(TK_ERROR_MODEL_LOAD_FAILED, ex.what()); }
    e->last_raw.assign((size_t)config->input_width * config->input_height, 0.0f);
    *out_estimator = e.release();
    return TK_SUCCESS;
}

void tk_depth_estimator_destroy(tk_depth_estimator_t** estimator) {
    if (!estimator || !*estimator) return;
    delete *estimator;
    *estimator = nullptr;
}

tk_error_code_t tk_depth_estimator_estimate(tk_depth_estimator_t* estimator, const tk_video_frame_t* video_frame, tk_vision_depth_map_t** out_depth_map) {
    if (!estimator || !video_frame || !out_depth_map) return TK_ERROR_INVALID_ARGUMENT;
    *out_depth_map = nullptr;
    if (!video_frame->data || video_frame->width == 0 || video_frame->height == 0) return dfail(TK_ERROR_INVALID_ARGUMENT, "empty frame");
    const uint32_t bpp = video_frame->format == TK_PIXEL_FORMAT_RGBA8 ? 4u : 3u;
    const uint32_t stride = video_frame->stride ? video_frame->stride : video_frame->width * bpp;
    if (stride < video_frame->width * bpp) return dfail(TK_ERROR_INVALID_ARGUMENT, "stride smaller than a row");
    tk_vision_depth_map_t* m = (tk_vision_depth_map_t*)malloc(sizeof(tk_vision_depth_map_t));
    if (!m) return TK_ERROR_OUT_OF_MEMORY;
    m->width = estimator->eng.width();
    m->height = estimator->eng.height();
    m->data = (float*)malloc((size_t)m->width * m->height * sizeof(float));
    if (!m->data) { free(m); return TK_ERROR_OUT_OF_MEMORY; }
    if (!estimator->eng.estimate(video_frame->data, video_frame->width, video_frame->height, stride, bpp, m->data, estimator->last_raw.data())) {
        free(m->data);
        free(m);
        return dfail(TK_ERROR_INFERENCE_FAILED, estimator->eng.error);
    }
    *out_depth_map = m;
    return TK_SUCCESS;
}

void tk_depth_estimator_free_map(tk_vision_depth_map_t** depth_map) {
    if (!depth_map || !*depth_map) return;
    free((*depth_map)->data);
    free(*depth_map);
    *depth_map = nullptr;
}

tk_error_code_t tk_mi355x_depth_onnx_probe(const char* path, int32_t* n_nodes) {
    if (!path) return TK_ERROR_INVALID_ARGUMENT;
    TkOnnxGraph g;
    try {
        if (!g.load(path)) return dfail(TK_ERROR_MODEL_LOAD_FAILED, g.error);
    } catch (const std::exception& ex) { return dfail(TK_ERROR_MODEL_LOAD_FAILED, ex.what()); }
    std::string err;
    if (!TkOnnxExec::ops_supported(g, &err)) return dfail(TK_ERROR_MODEL_VERIFICATION_FAILED, err);
    if (n_nodes) *n_nodes = (int32_t)g.nodes.size();
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_depth_forward_raw(tk_depth_estimator_t* estimator, const float* chw, float* raw_out, size_t raw_floats) {
    if (!estimator || !chw || !raw_out) return TK_ERROR_INVALID_ARGUMENT;
    if (raw_floats < (size_t)estimator->eng.width() * estimator->eng.height()) return dfail(TK_ERROR_INVALID_ARGUMENT, "raw_out too small");
    if (!estimator->eng.forward_raw(chw, raw_out)) return dfail(TK_ERROR_INFERENCE_FAILED, estimator->eng.error);
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_depth_last_raw(tk_depth_estimator_t* estimator, float* raw_out, size_t raw_floats) {
    if (!estimator || !raw_out) return TK_ERROR_INVALID_ARGUMENT;
    if (raw_floats < estimator->last_raw.size()) return dfail(TK_ERROR_INVALID_ARGUMENT, "raw_out too small");
    memcpy(raw_out, estimator->last_raw.data(), estimator->last_raw.size() * sizeof(float));
    return TK_SUCCESS;
}

tk_fused_result_t* tk_vision_rust_fuse_data(const tk_detection_result_t* detections, size_t detection_count, const tk_vision_depth_map_t* depth_map,
                                            uint32_t frame_width, uint32_t frame_height, float focal_length_x, float focal_length_y) {
    if (!detections || !depth_map) return nullptr; /* lib.rs:203-205 */
    std::vector<TkBox> boxes(detection_count);
    std::vector<uint32_t> cls(detection_count);
    for (size_t i = 0; i < detection_count; ++i) { boxes[i] = {detections[i].bbox.x, detections[i].bbox.y, detections[i].bbox.w, detections[i].bbox.h}; cls[i] = detections[i].class_id; }
    std::vector<TkFused> fused;
    std::vector<TkFusion::Tracker> after;
    {
        std::lock_guard<std::mutex> lk(g_fusion_mu);
        g_fusion.fuse(boxes.data(), cls.data(), detection_count, depth_map->data, depth_map->width, depth_map->height, frame_width, frame_height, focal_length_x,
                      focal_length_y, &fused);
        after = g_fusion.trackers();
    }
    size_t n = 0;
    for (const auto& f : fused) n += f.valid ? 1 : 0;
    tk_fused_result_t* r = (tk_fused_result_t*)calloc(1, sizeof(tk_fused_result_t));
    tk_enriched_object_t* objs = (tk_enriched_object_t*)calloc(n ? n : 1, sizeof(tk_enriched_object_t));
    if (!r || !objs) { free(r); free(objs); return nullptr; }
    size_t k = 0;
    for (size_t i = 0; i < detection_count; ++i) {
        if (!fused[i].valid) continue;
        tk_enriched_object_t& o = objs[k++];
        o.class_id = cls[i];
        for (const auto& t : after) if (t.id == fused[i].tracker_id) o.class_id = t.class_id; /* the tracker's class (object_analysis.rs:200) */
        o.confidence = 1.0f;
        o.bbox = detections[i].bbox;
        o.distance_meters = fused[i].distance_m;
        o.width_meters = fused[i].width_m;
        o.height_meters = fused[i].height_m;
        o.is_partially_occluded = false;
    }
    r->objects = objs;
    r->count = n;
    return r;
}

void tk_vision_rust_free_fused_result(tk_fused_result_t* result) {
    if (!result) return;
    free((void*)result->objects);
    free(result);
}

void tk_mi355x_fusion_reset(void) {
    std::lock_guard<std::mutex> lk(g_fusion_mu);
    g_fusion.clear();
}

float tk_mi355x_fusion_raw_distance(const tk_rect_t* bbox, const tk_vision_depth_map_t* depth_map, uint32_t frame_width, uint32_t frame_height) {
    if (!bbox || !depth_map || !depth_map->data || frame_width == 0 || frame_height == 0) return -1.0f;
    const TkBox b{bbox->x, bbox->y, bbox->w, bbox->h};
    return TkFusion::raw_distance(b, depth_map->data, depth_map->width, depth_map->height, frame_width, frame_height);
}

} /* extern "C" */
