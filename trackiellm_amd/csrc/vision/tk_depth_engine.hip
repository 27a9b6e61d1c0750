/* tk_depth_engine.hip — see tk_depth_engine.h */
#include "tk_depth_engine.h"

#include "../common/tk_exact_math.h"
#include "tk_vision_engine.h"

#define DQ(expr)                                                                                                  \
    do {                                                                                                          \
        hipError_t e__ = (expr);                                                                                  \
        if (e__ != hipSuccess) { error = std::string(#expr) + " failed: " + hipGetErrorString(e__); return false; } \
    } while (0)

/* min and max of the raw map: one workgroup, each thread scans a strided share, then a shuffle / LDS tree (min and max are
 * order-independent, so the result equals the reference's sequential scan, src/vision/tk_depth_midas.c:476-483) */
__global__ __launch_bounds__(1024) void k_depth_minmax(const float* x, int64_t n, float* mm) {
    __shared__ float smin[16], smax[16];
    float lo = x[0], hi = x[0];
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float v = x[i];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
    for (int s = 32; s >= 1; s >>= 1) {
        const float ol = __shfl_xor(lo, s, 64), oh = __shfl_xor(hi, s, 64);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) { lo = smin[w] < lo ? smin[w] : lo; hi = smax[w] > hi ? smax[w] : hi; }
        mm[0] = lo;
        mm[1] = hi;
    }
}

__global__ void k_depth_metric(const float* x, const float* mm, float min_depth, float max_depth, float* y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float lo = mm[0], hi = mm[1], range = hi - lo;
    if (range < 1e-6f) { y[i] = max_depth; return; }
    const float normalized = tk_divf(x[i] - lo, range);
    y[i] = max_depth - normalized * (max_depth - min_depth);
}

TkDepthEngine::~TkDepthEngine() {
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    exec_.unload();
    if (frame_dev_) (void)hipFree(frame_dev_);
    if (chw_dev_) (void)hipFree(chw_dev_);
    if (metric_dev_) (void)hipFree(metric_dev_);
    if (mm_dev_) (void)hipFree(mm_dev_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

bool TkDepthEngine::load(const char* onnx_path, int device, uint32_t in_w, uint32_t in_h) {
    device_ = device; in_w_ = in_w; in_h_ = in_h;
    DQ(hipSetDevice(device_));
    DQ(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    /* activations of one run are bump-allocated (no reuse inside a run): 2 GiB covers a MiDaS v2.1-class graph at 384 x 384 many times over */
    if (!exec_.load(onnx_path, device_, stream_, (size_t)1 << 29)) { error = exec_.error; return false; }
    const TkOnnxGraph& g = exec_.graph();
    for (const auto& vi : g.inputs)
        if (vi.elem_type == 1 || vi.elem_type == 0) { in_name_ = vi.name; /* the first float input (tk_depth_midas.c:295) */
            if (vi.dims.size() == 4) {
                const int64_t want[4] = {1, 3, (int64_t)in_h_, (int64_t)in_w_};
                for (int d = 0; d < 4; ++d)
                    if (vi.dims[(size_t)d] > 0 && vi.dims[(size_t)d] != want[d]) { error = "the model's input is not [1, 3, " + std::to_string(in_h_) + ", " + std::to_string(in_w_) + "]"; return false; }
            } else if (!vi.dims.empty()) { error = "the model's input must be 4-D (tk_depth_midas.c:306)"; return false; }
            break;
        }
    if (in_name_.empty()) { error = "the model has no float input"; return false; }
    if (g.outputs.empty()) { error = "the graph declares no outputs"; return false; }
    out_name_ = g.outputs[0].name;
    const size_t px = (size_t)in_w_ * in_h_;
    DQ(hipMalloc((void**)&chw_dev_, 3 * px * 4));
    DQ(hipMalloc((void**)&metric_dev_, px * 4));
    DQ(hipMalloc((void**)&mm_dev_, 2 * 4));
    /* a dry run on zeros finds unsupported attribute combinations and shape errors at load time, as session creation does */
    DQ(hipMemsetAsync(chw_dev_, 0, 3 * px * 4, stream_));
    const float* raw = nullptr;
    if (!run_network(&raw)) return false;
    DQ(hipStreamSynchronize(stream_));
    return true;
}

bool TkDepthEngine::run_network(const float** raw_dev) {
    exec_.begin();
    TkOnnxExec::Val in;
    in.d = chw_dev_;
    in.shape = {1, 3, (int64_t)in_h_, (int64_t)in_w_};
    exec_.bind(in_name_, in);
    if (!exec_.run()) { error = exec_.error; return false; }
    const TkOnnxExec::Val* out = exec_.value(out_name_);
    if (!out || !out->d || out->is_int) { error = "the graph did not produce its first output"; return false; }
    if (out->shape.size() < 3) { error = "the model's output must be [batch, height, width] or [batch, 1, height, width] (tk_depth_midas.c:321-324)"; return false; }
    if (out->count() != (int64_t)in_w_ * in_h_) {
        error = "the model's output holds " + std::to_string(out->count()) + " values, the depth map needs " + std::to_string((int64_t)in_w_ * in_h_);
        return false;
    }
    *raw_dev = out->d;
    return true;
}

bool TkDepthEngine::to_metric(const float* raw_dev, float* metric_dev, int64_t n) {
    hipLaunchKernelGGL(k_depth_minmax, dim3(1), dim3(1024), 0, stream_, raw_dev, n, mm_dev_);
    hipLaunchKernelGGL(k_depth_metric, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream_, raw_dev, mm_dev_, TK_DEPTH_MIN_M, TK_DEPTH_MAX_M, metric_dev, n);
    DQ(hipGetLastError());
    return true;
}

bool TkDepthEngine::estimate(const uint8_t* frame, uint32_t w, uint32_t h, uint32_t stride, uint32_t bpp, float* depth_out, float* raw_out) {
    DQ(hipSetDevice(device_));
    const size_t bytes = (size_t)stride * h;
    if (bytes > frame_cap_) {
        if (frame_dev_) (void)hipFree(frame_dev_);
        frame_dev_ = nullptr;
        DQ(hipMalloc((void**)&frame_dev_, bytes));
        frame_cap_ = bytes;
    }
    DQ(hipMemcpyAsync(frame_dev_, frame, bytes, hipMemcpyHostToDevice, stream_));
    TkPreprocessArgs a{};
    a.src = frame_dev_; a.in_w = w; a.in_h = h; a.in_stride = stride; a.bpp = bpp;
    a.dst = chw_dev_; a.out_w = in_w_; a.out_h = in_h_;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f}; /* tk_depth_midas.c:378-379 */
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean[c]; a.std_dev[c] = sd[c]; }
    a.nhwc = 0;
    tk_launch_preprocess(a, stream_);
    const float* raw = nullptr;
    if (!run_network(&raw)) return false;
    const int64_t n = (int64_t)in_w_ * in_h_;
    if (!to_metric(raw, metric_dev_, n)) return false;
    if (raw_out) DQ(hipMemcpyAsync(raw_out, raw, (size_t)n * 4, hipMemcpyDeviceToHost, stream_));
    DQ(hipMemcpyAsync(depth_out, metric_dev_, (size_t)n * 4, hipMemcpyDeviceToHost, stream_));
    DQ(hipStreamSynchronize(stream_));
    return true;
}

bool TkDepthEngine::forward_raw(const float* chw_host, float* raw_out) {
    DQ(hipSetDevice(device_));
    const size_t px = (size_t)in_w_ * in_h_;
    DQ(hipMemcpyAsync(chw_dev_, chw_host, 3 * px * 4, hipMemcpyHostToDevice, stream_));
    const float* raw = nullptr;
    if (!run_network(&raw)) return false;
    DQ(hipMemcpyAsync(raw_out, raw, px * 4, hipMemcpyDeviceToHost, stream_));
    DQ(hipStreamSynchronize(stream_));
    return true;
}
