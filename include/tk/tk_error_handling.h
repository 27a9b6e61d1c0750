/*
 * tk_error_handling.h — error codes of the tk_* C-ABI.
 * Restates (values identical) /root/reference/src/utils/tk_error_handling.h:40-124 and the
 * thread-local detail string API (:185, :201).
 *
 * Codes the reference's hot-path sources use but its enum never declares are mapped to the
 * nearest declared value (SURVEY.md §8b "Errors"):
 *   TK_ERROR_MODEL_INFERENCE_FAILED -> TK_ERROR_INFERENCE_FAILED (4002)
 *   TK_ERROR_INTERNAL / _SYSTEM_ERROR / _CRITICAL_FAILURE -> TK_ERROR_UNKNOWN (1000)
 *   TK_ERROR_GPU_KERNEL_LAUNCH_FAILED -> TK_ERROR_GPU_KERNEL_LAUNCH (5006)
 *   TK_ERROR_INVALID_DIMENSIONS -> TK_ERROR_INVALID_ARGUMENT (1001)
 *   TK_ERROR_RESOURCE_EXHAUSTED -> TK_ERROR_OUT_OF_MEMORY (2000)
 */
#ifndef TK_MI355X_ERROR_HANDLING_H
#define TK_MI355X_ERROR_HANDLING_H

#if defined(__GNUC__) || defined(__clang__)
#define TK_NODISCARD __attribute__((warn_unused_result))
#else
#define TK_NODISCARD
#endif
#define TK_API __attribute__((visibility("default")))

typedef enum tk_error_code_t {
    TK_SUCCESS = 0,
    TK_ERROR_UNKNOWN = 1000,
    TK_ERROR_INVALID_ARGUMENT,
    TK_ERROR_INVALID_STATE,
    TK_ERROR_NOT_IMPLEMENTED,
    TK_ERROR_BUFFER_TOO_SMALL,
    TK_ERROR_TIMEOUT,
    TK_ERROR_PERMISSION_DENIED,
    TK_ERROR_NOT_INITIALIZED,
    TK_ERROR_OUT_OF_MEMORY = 2000,
    TK_ERROR_MEMORY_ALIGNMENT,
    TK_ERROR_MEMORY_POOL_EXHAUSTED,
    TK_ERROR_MEMORY_DOUBLE_FREE,
    TK_ERROR_MEMORY_INVALID_POINTER,
    TK_ERROR_IO = 3000,
    TK_ERROR_FILE_NOT_FOUND,
    TK_ERROR_FILE_READ,
    TK_ERROR_FILE_WRITE,
    TK_ERROR_FILE_CORRUPT,
    TK_ERROR_CONFIG_PARSE_FAILED,
    TK_ERROR_MODEL_LOAD_FAILED = 4000,
    TK_ERROR_MODEL_VERIFICATION_FAILED,
    TK_ERROR_INFERENCE_FAILED,
    TK_ERROR_INVALID_INPUT_TENSOR,
    TK_ERROR_INVALID_OUTPUT_TENSOR,
    TK_ERROR_BACKEND_NOT_SUPPORTED,
    TK_ERROR_GPU_ERROR = 5000,
    TK_ERROR_GPU_DEVICE_NOT_FOUND,
    TK_ERROR_GPU_DRIVER_VERSION,
    TK_ERROR_GPU_CUDA_ERROR,
    TK_ERROR_GPU_METAL_ERROR,
    TK_ERROR_GPU_ROCM_ERROR,
    TK_ERROR_GPU_KERNEL_LAUNCH,
    TK_ERROR_GPU_MEMORY,
    TK_ERROR_NETWORK_ERROR = 6000,
    TK_ERROR_CONNECTION_FAILED,
    TK_ERROR_CONNECTION_CLOSED,
    TK_ERROR_DNS_RESOLUTION_FAILED,
    TK_ERROR_SOCKET_ERROR,
    TK_ERROR_THREAD_CREATE_FAILED = 7000,
    TK_ERROR_MUTEX_ERROR,
    TK_ERROR_SEMAPHORE_ERROR,
    TK_ERROR_TASK_QUEUE_FULL,
    TK_ERROR_FUTURE_CANCELLED,
    TK_ERROR_FFI_PANIC = 8000,
    TK_ERROR_FFI_INVALID_STRING,
    TK_ERROR_CODE_COUNT
} tk_error_code_t;

/* TK_ERROR_INVALID_FORMAT is returned by the reference's LLM-response parser (src/cortex/tk_decision_engine.c:1644,1706,...) and declared
 * nowhere: it maps to the declared parse-failure code */
#define TK_ERROR_INVALID_FORMAT TK_ERROR_CONFIG_PARSE_FAILED

#ifdef __cplusplus
extern "C" {
#endif

/* reference: utils/tk_error_handling.h:185 / :201 — printf-style, thread-local */
TK_API void tk_error_set_detail(const char* fmt, ...);
TK_API const char* tk_error_get_detail(void);
/* reference: utils/tk_error_handling.h (tk_error_to_string) */
TK_API const char* tk_error_to_string(tk_error_code_t code);

#ifdef __cplusplus
}
#endif
#endif
