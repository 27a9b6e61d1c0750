/*
 * tk_depth_engine.h — monocular depth on the GPU: frame -> pre-process -> the depth network's ONNX graph -> metric depth map.
 *
 * Mirrors src/vision/tk_depth_midas.c: preprocess_frame (:374-395: the shared bilinear stretch-resize with the ImageNet mean / std,
 * planar CHW), run_inference (:397-440: one ORT Run, first input, first output), postprocess_depth + convert_inverse_depth_to_metric
 * (:442-499: min / max over the raw map, normalised = (d - min) / (max - min), depth = 10 - normalised * (10 - 0.1); a flat map
 * (max - min < 1e-6) becomes 10 m everywhere).  The network runs through csrc/nn/tk_onnx_exec (no ONNX Runtime).
 */
#ifndef TK_DEPTH_ENGINE_H
#define TK_DEPTH_ENGINE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../nn/tk_onnx_exec.h"

#define TK_DEPTH_MIN_M 0.1f  /* src/vision/tk_depth_midas.c:46-47 */
#define TK_DEPTH_MAX_M 10.0f

class TkDepthEngine {
public:
    std::string error;
    ~TkDepthEngine();
    bool load(const char* onnx_path, int device, uint32_t in_w, uint32_t in_h);
    /* host frame (RGB8 / RGBA8, stride in bytes) -> metric depth [in_h][in_w] on the host; raw_out (optional) receives the network's
     * output before the metric conversion */
    bool estimate(const uint8_t* frame, uint32_t w, uint32_t h, uint32_t stride, uint32_t bpp, float* depth_out, float* raw_out);
    /* test hook: the network alone on a pre-processed planar tensor [3][in_h][in_w] */
    bool forward_raw(const float* chw_host, float* raw_out);
    /* the metric conversion alone (device pointers), as tk_kernels_postprocess_depth_map's per-map companion */
    bool to_metric(const float* raw_dev, float* metric_dev, int64_t n);
    uint32_t width() const { return in_w_; }
    uint32_t height() const { return in_h_; }
    int node_count() const { return (int)exec_.graph().nodes.size(); }
    hipStream_t stream() const { return stream_; }

private:
    bool run_network(const float** raw_dev);
    TkOnnxExec exec_;
    int device_ = 0;
    uint32_t in_w_ = 0, in_h_ = 0;
    hipStream_t stream_ = nullptr;
    std::string in_name_, out_name_;
    uint8_t* frame_dev_ = nullptr;
    size_t frame_cap_ = 0;
    float *chw_dev_ = nullptr, *metric_dev_ = nullptr, *mm_dev_ = nullptr;
};

#endif
