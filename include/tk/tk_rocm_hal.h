/*
 * tk_rocm_hal.h — the reference's thin GPU C-ABI layer ("HAL"), ROCm flavour.
 *   parameter structs + kernel launchers   src/gpu/rocm/tk_rocm_kernels.hpp:58-76,96 (preprocess), :106-133 (depth post-process),
 *                                          :143-176 (depth -> point cloud); generic twin src/gpu/tk_gpu_helper.h:42-57
 *   dispatcher                             src/gpu/rocm/tk_rocm_dispatch.hpp:42-217
 * Differences kept deliberately small and listed in ABI_NOTES.md:
 *   - tk_kernels_preprocess_image samples with the reference CPU pre-processor's formula (the canonical oracle,
 *     SURVEY.md §0 F5), not the top-left mapping of src/gpu/rocm/tk_rocm_kernels.cpp:70-71; scale == 1/255 is applied
 *     as the CPU path's division by 255, any other scale as a multiplication;
 *   - the dispatcher keeps one persistent context per device (streams, no per-call handle creation — the reference
 *     creates rocBLAS/MIOpen handles per call, src/gpu/extensions/rocm/tk_rocm_tensor_ops.cpp:206-207,250-251).
 * hipStream_t is spelled `void*` here so plain C hosts need no HIP headers.
 */
#ifndef TK_MI355X_ROCM_HAL_H
#define TK_MI355X_ROCM_HAL_H

#include "tk_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } tk_float3;
typedef void* tk_hip_stream_t;

typedef struct {
    const unsigned char* d_input_image;
    uint32_t input_width;
    uint32_t input_height;
    uint32_t input_stride_bytes;
    float* d_output_tensor;
    uint32_t output_width;
    uint32_t output_height;
    tk_float3 mean;
    tk_float3 std_dev;
    float scale;
} tk_preprocess_params_t;

typedef struct {
    const float* d_raw_depth_map;
    uint32_t width;
    uint32_t height;
    float* d_metric_depth_map;
    float scale;
    float shift;
} tk_postprocess_depth_params_t;

typedef struct {
    const float* d_metric_depth_map;
    uint32_t width;
    uint32_t height;
    tk_float3* d_point_cloud;
    float fx, fy, cx, cy;
} tk_depth_to_points_params_t;

/* src/gpu/tk_gpu_helper.h:42-57 — the form the only reference call site fills (src/vision/tk_object_detector.c:231-243) */
typedef struct { float x, y, z; } TkFloat3;
typedef struct {
    const void* d_input_image;
    uint32_t input_width;
    uint32_t input_height;
    uint32_t input_stride_bytes;
    void* d_output_tensor;
    uint32_t output_width;
    uint32_t output_height;
    TkFloat3 mean;
    TkFloat3 std_dev;
} tk_preprocess_params_generic_t;

TK_API TK_NODISCARD tk_error_code_t tk_kernels_preprocess_image(const tk_preprocess_params_t* params, tk_hip_stream_t stream);
TK_API TK_NODISCARD tk_error_code_t tk_kernels_preprocess_image_generic(const tk_preprocess_params_generic_t* params, tk_hip_stream_t stream);
TK_API TK_NODISCARD tk_error_code_t tk_kernels_postprocess_depth_map(const tk_postprocess_depth_params_t* params, tk_hip_stream_t stream);
TK_API TK_NODISCARD tk_error_code_t tk_kernels_depth_to_point_cloud(const tk_depth_to_points_params_t* params, tk_hip_stream_t stream);

/* src/gpu/tk_gpu_helper.h:113-130 + src/gpu/cuda/tk_cuda_kernels.h:92 — row softmax of a [num_rows][num_cols] f32 tensor.  The reference
 * ships it for CUDA only (tk_cuda_kernels.cu:312-393: power-of-two num_cols <= 1024); this one takes any num_cols and in-place calls
 * (d_output_tensor == d_input_tensor).  It is the kernel the Whisper attention rows go through, and the one numeric known-answer test the
 * reference holds (tests/tk_gpu_softmax_test.cpp) is replayed against it. */
typedef struct {
    const void* d_input_tensor;
    void* d_output_tensor;
    uint32_t num_rows;
    uint32_t num_cols;
} tk_softmax_params_t;
TK_API TK_NODISCARD tk_error_code_t tk_kernels_softmax(const tk_softmax_params_t* params, tk_hip_stream_t stream);

typedef struct tk_rocm_dispatcher_s tk_rocm_dispatcher_t;
typedef struct tk_gpu_buffer_s* tk_gpu_buffer_t;
typedef struct { int device_id; } tk_rocm_dispatcher_config_t;

TK_API TK_NODISCARD tk_error_code_t tk_rocm_dispatch_create(tk_rocm_dispatcher_t** out_dispatcher, const tk_rocm_dispatcher_config_t* config);
TK_API void tk_rocm_dispatch_destroy(tk_rocm_dispatcher_t** dispatcher);
TK_API TK_NODISCARD tk_error_code_t tk_rocm_dispatch_malloc(tk_rocm_dispatcher_t* dispatcher, tk_gpu_buffer_t* out_buffer, size_t size_bytes);
TK_API void tk_rocm_dispatch_free(tk_rocm_dispatcher_t* dispatcher, tk_gpu_buffer_t* buffer);
TK_API TK_NODISCARD tk_error_code_t tk_rocm_dispatch_upload_async(tk_rocm_dispatcher_t* dispatcher, tk_gpu_buffer_t dst_buffer, const void* src_host_ptr,
                                                                  size_t size_bytes);
TK_API TK_NODISCARD tk_error_code_t tk_rocm_dispatch_download_async(tk_rocm_dispatcher_t* dispatcher, void* dst_host_ptr, tk_gpu_buffer_t src_buffer,
                                                                    size_t size_bytes);
TK_API TK_NODISCARD tk_error_code_t tk_rocm_dispatch_synchronize(tk_rocm_dispatcher_t* dispatcher);
TK_API TK_NODISCARD tk_error_code_t tk_rocm_dispatch_get_stream(tk_rocm_dispatcher_t* dispatcher, tk_hip_stream_t* stream);
TK_API TK_NODISCARD tk_error_code_t tk_rocm_dispatch_preprocess_image(tk_rocm_dispatcher_t* dispatcher, const tk_preprocess_params_t* params);
TK_API TK_NODISCARD tk_error_code_t tk_rocm_dispatch_depth_to_point_cloud(tk_rocm_dispatcher_t* dispatcher, const tk_depth_to_points_params_t* params);
/* device address behind an opaque buffer (what a caller puts into the *_params_t structs) */
TK_API void* tk_rocm_dispatch_buffer_ptr(tk_gpu_buffer_t buffer);

#ifdef __cplusplus
}
#endif
#endif
