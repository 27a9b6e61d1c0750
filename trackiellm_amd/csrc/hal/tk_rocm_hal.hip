/*
 * tk_rocm_hal.hip — dispatcher + kernel launchers of the reference's ROCm HAL surface
 * (src/gpu/rocm/tk_rocm_dispatch.hpp:83-217, src/gpu/rocm/tk_rocm_kernels.hpp:96,133,176).
 */
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <vector>

#include "../common/tk_exact_math.h"
#include "tk/tk_mi355x_ext.h"

#include "../vision/tk_vision_engine.h"
#include "tk/tk_rocm_hal.h"
#include "../nn/tk_gemm_tiled.h"
#include "../nn/tk_nn_kernels.h"

struct tk_gpu_buffer_s {
    void* dptr;
    size_t size;
};

struct tk_rocm_dispatcher_s {
    int device;
    hipStream_t compute, h2d, d2h; /* three non-blocking streams, as the reference dispatcher */
    hipEvent_t ev_upload, ev_compute;
};

/* out = raw * scale + shift  (src/gpu/rocm/tk_rocm_kernels.cpp:148-163), grid-stride, float4-free: the maps are tiny */
__global__ void k_postprocess_depth(tk_postprocess_depth_params_t p) {
    const size_t n = (size_t)p.width * p.height;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p.d_metric_depth_map[i] = p.d_raw_depth_map[i] * p.scale + p.shift;
}

/* pinhole unprojection (src/gpu/rocm/tk_rocm_kernels.cpp:176-202): depth <= 0 -> origin */
__global__ void k_depth_to_points(tk_depth_to_points_params_t p) {
    const size_t n = (size_t)p.width * p.height;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = p.d_metric_depth_map[i];
        tk_float3 o = {0.0f, 0.0f, 0.0f};
        if (d > 0.0f) {
            const float u = (float)(i % p.width), v = (float)(i / p.width);
            o.x = tk_divf((u - p.cx) * d, p.fx);
            o.y = tk_divf((v - p.cy) * d, p.fy);
            o.z = d;
        }
        p.d_point_cloud[i] = o;
    }
}

/* multiply-by-scale variant of the canonical pre-processor for scale != 1/255 */
__global__ void k_preprocess_scaled(TkPreprocessArgs a, float scale) {
    const uint32_t ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y * blockDim.y + threadIdx.y;
    if (ox >= a.out_w || oy >= a.out_h) return;
    const float x_ratio = tk_divf((float)a.in_w - 1.0f, (float)a.out_w), y_ratio = tk_divf((float)a.in_h - 1.0f, (float)a.out_h);
    const float gx = x_ratio * (float)ox, gy = y_ratio * (float)oy;
    const int x = (int)gx, y = (int)gy;
    const float xd = gx - (float)x, yd = gy - (float)y;
    const int x1 = x + 1 < (int)a.in_w ? x + 1 : x, y1 = y + 1 < (int)a.in_h ? y + 1 : y;
    const uint8_t* r0 = a.src + (size_t)y * a.in_stride;
    const uint8_t* r1 = a.src + (size_t)y1 * a.in_stride;
    const size_t np = (size_t)a.out_w * a.out_h, pix = (size_t)oy * a.out_w + ox;
    for (int c = 0; c < 3; ++c) {
        float v = ((float)r0[x * a.bpp + c] * (1.0f - xd)) * (1.0f - yd);
        v = v + ((float)r0[x1 * a.bpp + c] * xd) * (1.0f - yd);
        v = v + ((float)r1[x * a.bpp + c] * (1.0f - xd)) * yd;
        v = v + ((float)r1[x1 * a.bpp + c] * xd) * yd;
        a.dst[c * np + pix] = tk_divf(v * scale - a.mean[c], a.std_dev[c]);
    }
}

static tk_error_code_t launch_status() { return hipGetLastError() == hipSuccess ? TK_SUCCESS : TK_ERROR_GPU_KERNEL_LAUNCH; }

extern "C" {

tk_error_code_t tk_kernels_preprocess_image(const tk_preprocess_params_t* p, tk_hip_stream_t stream) {
    if (!p || !p->d_input_image || !p->d_output_tensor || p->output_width == 0 || p->output_height == 0) return TK_ERROR_INVALID_ARGUMENT;
    if (p->input_width < 2 || p->input_height < 2 || p->input_stride_bytes < p->input_width * 3) return TK_ERROR_INVALID_ARGUMENT;
    TkPreprocessArgs a{};
    a.src = p->d_input_image; a.in_w = p->input_width; a.in_h = p->input_height; a.in_stride = p->input_stride_bytes; a.bpp = 3;
    a.dst = p->d_output_tensor; a.out_w = p->output_width; a.out_h = p->output_height; a.nhwc = 0;
    a.mean[0] = p->mean.x; a.mean[1] = p->mean.y; a.mean[2] = p->mean.z;
    a.std_dev[0] = p->std_dev.x; a.std_dev[1] = p->std_dev.y; a.std_dev[2] = p->std_dev.z;
    if (p->scale == 1.0f / 255.0f || p->scale == 0.0f) tk_launch_preprocess(a, (hipStream_t)stream);
    else hipLaunchKernelGGL(k_preprocess_scaled, dim3((a.out_w + 63) / 64, (a.out_h + 3) / 4), dim3(64, 4), 0, (hipStream_t)stream, a, p->scale);
    return launch_status();
}

tk_error_code_t tk_kernels_preprocess_image_generic(const tk_preprocess_params_generic_t* g, tk_hip_stream_t stream) {
    if (!g) return TK_ERROR_INVALID_ARGUMENT;
    tk_preprocess_params_t p;
    p.d_input_image = (const unsigned char*)g->d_input_image; p.input_width = g->input_width; p.input_height = g->input_height;
    p.input_stride_bytes = g->input_stride_bytes; p.d_output_tensor = (float*)g->d_output_tensor; p.output_width = g->output_width;
    p.output_height = g->output_height; p.mean = tk_float3{g->mean.x, g->mean.y, g->mean.z}; p.std_dev = tk_float3{g->std_dev.x, g->std_dev.y, g->std_dev.z};
    p.scale = 1.0f / 255.0f;
    return tk_kernels_preprocess_image(&p, stream);
}

tk_error_code_t tk_kernels_postprocess_depth_map(const tk_postprocess_depth_params_t* p, tk_hip_stream_t stream) {
    if (!p || !p->d_raw_depth_map || !p->d_metric_depth_map || p->width == 0 || p->height == 0) return TK_ERROR_INVALID_ARGUMENT;
    const size_t n = (size_t)p->width * p->height;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_postprocess_depth, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *p);
    return launch_status();
}

tk_error_code_t tk_kernels_softmax(const tk_softmax_params_t* p, tk_hip_stream_t stream) {
    if (!p || !p->d_input_tensor || !p->d_output_tensor || p->num_rows == 0 || p->num_cols == 0) return TK_ERROR_INVALID_ARGUMENT;
    if (p->d_output_tensor != p->d_input_tensor &&
        hipMemcpyAsync(p->d_output_tensor, p->d_input_tensor, (size_t)p->num_rows * p->num_cols * sizeof(float), hipMemcpyDeviceToDevice,
                       (hipStream_t)stream) != hipSuccess)
        return TK_ERROR_GPU_ROCM_ERROR;
    tk_launch_softmax_rows((float*)p->d_output_tensor, (int)p->num_rows, (int)p->num_cols, (int)p->num_cols, (hipStream_t)stream);
    return launch_status();
}

tk_error_code_t tk_mi355x_gemm_pair(int device, int M, int N, int K, const float* a, const float* w, const float* bias, const float* residual, int act,
                                    int f16, float* c_staged, float* c_tiled) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 128 || !a || !w || !c_staged || !c_tiled || act < 0 || act > 3) return TK_ERROR_INVALID_ARGUMENT;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return TK_ERROR_GPU_DEVICE_NOT_FOUND;
    if (device < 0 || device >= n || hipSetDevice(device) != hipSuccess) return TK_ERROR_INVALID_ARGUMENT;
    if (!tk_nn_prepare_device() || !tk_gemm_tiled_prepare_device()) return TK_ERROR_GPU_ROCM_ERROR;
    const size_t na = (size_t)M * K, nw = (size_t)N * K, nc = (size_t)M * N;
    const int64_t blk = M >= 256 ? 256 : (M > 128 ? 256 : M > 64 ? 128 : M > 32 ? 64 : M > 16 ? 32 : 16);
    const size_t nimg = (size_t)(((int64_t)M + blk - 1) / blk * blk) * K;
    float *da = nullptr, *dw = nullptr, *db = nullptr, *dr = nullptr, *dc = nullptr, *dimg = nullptr;
    uint16_t* dwh = nullptr;
    uint8_t* dt = nullptr;
    bool ok = hipMalloc((void**)&da, na * 4) == hipSuccess && hipMalloc((void**)&dw, nw * 4) == hipSuccess && hipMalloc((void**)&dc, nc * 4) == hipSuccess &&
              hipMalloc((void**)&dimg, nimg * 4) == hipSuccess && hipMalloc((void**)&dt, tk_tiled_weight_bytes(N, K, f16 ? 2 : 4)) == hipSuccess &&
              (!bias || hipMalloc((void**)&db, (size_t)N * 4) == hipSuccess) && (!residual || hipMalloc((void**)&dr, nc * 4) == hipSuccess) &&
              (!f16 || hipMalloc((void**)&dwh, nw * 2) == hipSuccess);
    ok = ok && hipMemcpy(da, a, na * 4, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dw, w, nw * 4, hipMemcpyHostToDevice) == hipSuccess &&
         (!bias || hipMemcpy(db, bias, (size_t)N * 4, hipMemcpyHostToDevice) == hipSuccess) &&
         (!residual || hipMemcpy(dr, residual, nc * 4, hipMemcpyHostToDevice) == hipSuccess) && hipMemset(dimg, 0, nimg * 4) == hipSuccess;
    if (ok) {
        std::vector<uint16_t> wh;
        if (f16) { /* the f16 checkpoint's weights; the staged kernel reads the same values through its f16 B operand */
            wh.resize(nw);
            for (size_t i = 0; i < nw; ++i) wh[i] = tk_f32_to_f16(w[i]);
            ok = hipMemcpy(dwh, wh.data(), nw * 2, hipMemcpyHostToDevice) == hipSuccess;
        }
        /* staged kernel: with f16 the activations must already be f16-rounded values, as the LLM's producers write them */
        std::vector<float> ar;
        if (ok && f16) {
            ar.resize(na);
            for (size_t i = 0; i < na; ++i) ar[i] = tk_f16_to_f32(tk_f32_to_f16(a[i]));
            ok = hipMemcpy(da, ar.data(), na * 4, hipMemcpyHostToDevice) == hipSuccess;
        }
        if (ok) {
            TkGemm g{};
            g.A = da; g.B = f16 ? (const float*)dwh : dw; g.C = dc; g.bias = db; g.residual = dr;
            g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.ldr = N; g.act = act; g.alpha = 1.0f; g.batch = 1; g.b_f16 = f16 ? 1 : 0;
            tk_launch_gemm(g, nullptr);
            ok = hipMemcpy(c_staged, dc, nc * 4, hipMemcpyDeviceToHost) == hipSuccess && hipMemset(dc, 0, nc * 4) == hipSuccess;
        }
        if (ok) {
            tk_launch_tile_weights(f16 ? (const void*)dwh : (const void*)dw, f16 ? 2 : 4, N, K, dt, nullptr);
            tk_launch_pack_a(da, M, K, K, 0, dimg, nullptr); /* da already holds the f16-rounded values in the f16 case */
            TkTiledGemm t{};
            t.tiles[0] = dt; t.row_tiles[0] = (N + 15) / 16; t.nseg = 1; t.wbytes = f16 ? 2 : 4;
            t.K = K; t.ks = 1; t.ldc = N; t.n_valid = N; t.nrows = M; t.a_img = dimg; t.a_ts = (size_t)K * 16; t.out = dc;
            t.bias = db; t.residual = dr; t.ldr = N; t.act = act; t.add_zero_bias = 1;
            ok = tk_launch_gemm_tiled(t, nullptr) && hipGetLastError() == hipSuccess && hipMemcpy(c_tiled, dc, nc * 4, hipMemcpyDeviceToHost) == hipSuccess;
        }
    }
    void* ptrs[] = {da, dw, db, dr, dc, dimg, dwh, dt};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    return ok ? TK_SUCCESS : TK_ERROR_GPU_ROCM_ERROR;
}

tk_error_code_t tk_kernels_depth_to_point_cloud(const tk_depth_to_points_params_t* p, tk_hip_stream_t stream) {
    if (!p || !p->d_metric_depth_map || !p->d_point_cloud || p->width == 0 || p->height == 0 || p->fx == 0.0f || p->fy == 0.0f)
        return TK_ERROR_INVALID_ARGUMENT;
    const size_t n = (size_t)p->width * p->height;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_depth_to_points, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *p);
    return launch_status();
}

tk_error_code_t tk_rocm_dispatch_create(tk_rocm_dispatcher_t** out, const tk_rocm_dispatcher_config_t* config) {
    if (!out || !config) return TK_ERROR_INVALID_ARGUMENT;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return TK_ERROR_GPU_DEVICE_NOT_FOUND;
    if (config->device_id < 0 || config->device_id >= n) return TK_ERROR_INVALID_ARGUMENT;
    tk_rocm_dispatcher_s* d = (tk_rocm_dispatcher_s*)calloc(1, sizeof *d);
    if (!d) return TK_ERROR_OUT_OF_MEMORY;
    d->device = config->device_id;
    if (hipSetDevice(d->device) != hipSuccess || hipStreamCreateWithFlags(&d->compute, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&d->h2d, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&d->d2h, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&d->ev_upload, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&d->ev_compute, hipEventDisableTiming) != hipSuccess) {
        free(d);
        return TK_ERROR_GPU_ROCM_ERROR;
    }
    *out = d;
    return TK_SUCCESS;
}

void tk_rocm_dispatch_destroy(tk_rocm_dispatcher_t** dispatcher) {
    if (!dispatcher || !*dispatcher) return;
    tk_rocm_dispatcher_s* d = *dispatcher;
    (void)hipSetDevice(d->device);
    (void)hipDeviceSynchronize();
    (void)hipStreamDestroy(d->compute); (void)hipStreamDestroy(d->h2d); (void)hipStreamDestroy(d->d2h);
    (void)hipEventDestroy(d->ev_upload); (void)hipEventDestroy(d->ev_compute);
    free(d);
    *dispatcher = NULL;
}

tk_error_code_t tk_rocm_dispatch_malloc(tk_rocm_dispatcher_t* d, tk_gpu_buffer_t* out_buffer, size_t size_bytes) {
    if (!d || !out_buffer || size_bytes == 0) return TK_ERROR_INVALID_ARGUMENT;
    if (hipSetDevice(d->device) != hipSuccess) return TK_ERROR_GPU_ROCM_ERROR;
    tk_gpu_buffer_s* b = (tk_gpu_buffer_s*)calloc(1, sizeof *b);
    if (!b) return TK_ERROR_OUT_OF_MEMORY;
    if (hipMalloc(&b->dptr, size_bytes) != hipSuccess) { free(b); return TK_ERROR_GPU_MEMORY; }
    b->size = size_bytes;
    *out_buffer = b;
    return TK_SUCCESS;
}

void tk_rocm_dispatch_free(tk_rocm_dispatcher_t* d, tk_gpu_buffer_t* buffer) {
    if (!d || !buffer || !*buffer) return;
    (void)hipSetDevice(d->device);
    (void)hipFree((*buffer)->dptr);
    free(*buffer);
    *buffer = NULL;
}

void* tk_rocm_dispatch_buffer_ptr(tk_gpu_buffer_t buffer) { return buffer ? buffer->dptr : NULL; }

tk_error_code_t tk_rocm_dispatch_upload_async(tk_rocm_dispatcher_t* d, tk_gpu_buffer_t dst, const void* src, size_t n) {
    if (!d || !dst || !src || n > dst->size) return TK_ERROR_INVALID_ARGUMENT;
    if (hipSetDevice(d->device) != hipSuccess) return TK_ERROR_GPU_ROCM_ERROR;
    if (hipMemcpyAsync(dst->dptr, src, n, hipMemcpyHostToDevice, d->h2d) != hipSuccess) return TK_ERROR_GPU_MEMORY;
    /* kernels enqueued afterwards must see the data */
    if (hipEventRecord(d->ev_upload, d->h2d) != hipSuccess || hipStreamWaitEvent(d->compute, d->ev_upload, 0) != hipSuccess) return TK_ERROR_GPU_ROCM_ERROR;
    return TK_SUCCESS;
}

tk_error_code_t tk_rocm_dispatch_download_async(tk_rocm_dispatcher_t* d, void* dst, tk_gpu_buffer_t src, size_t n) {
    if (!d || !dst || !src || n > src->size) return TK_ERROR_INVALID_ARGUMENT;
    if (hipSetDevice(d->device) != hipSuccess) return TK_ERROR_GPU_ROCM_ERROR;
    if (hipEventRecord(d->ev_compute, d->compute) != hipSuccess || hipStreamWaitEvent(d->d2h, d->ev_compute, 0) != hipSuccess) return TK_ERROR_GPU_ROCM_ERROR;
    if (hipMemcpyAsync(dst, src->dptr, n, hipMemcpyDeviceToHost, d->d2h) != hipSuccess) return TK_ERROR_GPU_MEMORY;
    return TK_SUCCESS;
}

tk_error_code_t tk_rocm_dispatch_synchronize(tk_rocm_dispatcher_t* d) {
    if (!d) return TK_ERROR_INVALID_ARGUMENT;
    if (hipSetDevice(d->device) != hipSuccess) return TK_ERROR_GPU_ROCM_ERROR;
    if (hipStreamSynchronize(d->h2d) != hipSuccess || hipStreamSynchronize(d->compute) != hipSuccess || hipStreamSynchronize(d->d2h) != hipSuccess)
        return TK_ERROR_GPU_ROCM_ERROR;
    return TK_SUCCESS;
}

tk_error_code_t tk_rocm_dispatch_get_stream(tk_rocm_dispatcher_t* d, tk_hip_stream_t* stream) {
    if (!d || !stream) return TK_ERROR_INVALID_ARGUMENT;
    *stream = (tk_hip_stream_t)d->compute;
    return TK_SUCCESS;
}

tk_error_code_t tk_rocm_dispatch_preprocess_image(tk_rocm_dispatcher_t* d, const tk_preprocess_params_t* params) {
    if (!d) return TK_ERROR_INVALID_ARGUMENT;
    if (hipSetDevice(d->device) != hipSuccess) return TK_ERROR_GPU_ROCM_ERROR;
    return tk_kernels_preprocess_image(params, (tk_hip_stream_t)d->compute);
}

tk_error_code_t tk_rocm_dispatch_depth_to_point_cloud(tk_rocm_dispatcher_t* d, const tk_depth_to_points_params_t* params) {
    if (!d) return TK_ERROR_INVALID_ARGUMENT;
    if (hipSetDevice(d->device) != hipSuccess) return TK_ERROR_GPU_ROCM_ERROR;
    return tk_kernels_depth_to_point_cloud(params, (tk_hip_stream_t)d->compute);
}

} /* extern "C" */
