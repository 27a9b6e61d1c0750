/*
 * tk_abi_common.cpp — error detail strings, tk_path_t helpers, version.
 * Mirrors src/utils/tk_error_handling.c (thread-local detail buffer) and
 * src/internal_tools/tk_file_manager.h:119,172.
 */
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tk/tk_mi355x_ext.h"
#include "tk/tk_types.h"

static thread_local char g_detail[1024] = {0};

extern "C" {

void tk_error_set_detail(const char* fmt, ...) {
    if (!fmt) { g_detail[0] = 0; return; }
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_detail, sizeof g_detail, fmt, ap);
    va_end(ap);
}

const char* tk_error_get_detail(void) { return g_detail; }

const char* tk_error_to_string(tk_error_code_t code) {
    switch (code) {
        case TK_SUCCESS: return "TK_SUCCESS";
        case TK_ERROR_UNKNOWN: return "TK_ERROR_UNKNOWN";
        case TK_ERROR_INVALID_ARGUMENT: return "TK_ERROR_INVALID_ARGUMENT";
        case TK_ERROR_INVALID_STATE: return "TK_ERROR_INVALID_STATE";
        case TK_ERROR_NOT_IMPLEMENTED: return "TK_ERROR_NOT_IMPLEMENTED";
        case TK_ERROR_BUFFER_TOO_SMALL: return "TK_ERROR_BUFFER_TOO_SMALL";
        case TK_ERROR_NOT_INITIALIZED: return "TK_ERROR_NOT_INITIALIZED";
        case TK_ERROR_OUT_OF_MEMORY: return "TK_ERROR_OUT_OF_MEMORY";
        case TK_ERROR_FILE_NOT_FOUND: return "TK_ERROR_FILE_NOT_FOUND";
        case TK_ERROR_FILE_CORRUPT: return "TK_ERROR_FILE_CORRUPT";
        case TK_ERROR_MODEL_LOAD_FAILED: return "TK_ERROR_MODEL_LOAD_FAILED";
        case TK_ERROR_INFERENCE_FAILED: return "TK_ERROR_INFERENCE_FAILED";
        case TK_ERROR_INVALID_INPUT_TENSOR: return "TK_ERROR_INVALID_INPUT_TENSOR";
        case TK_ERROR_BACKEND_NOT_SUPPORTED: return "TK_ERROR_BACKEND_NOT_SUPPORTED";
        case TK_ERROR_GPU_ERROR: return "TK_ERROR_GPU_ERROR";
        case TK_ERROR_GPU_DEVICE_NOT_FOUND: return "TK_ERROR_GPU_DEVICE_NOT_FOUND";
        case TK_ERROR_GPU_ROCM_ERROR: return "TK_ERROR_GPU_ROCM_ERROR";
        case TK_ERROR_GPU_KERNEL_LAUNCH: return "TK_ERROR_GPU_KERNEL_LAUNCH";
        case TK_ERROR_GPU_MEMORY: return "TK_ERROR_GPU_MEMORY";
        default: return "TK_ERROR";
    }
}

tk_error_code_t tk_path_create_from_string(tk_path_t** out_path, const char* path_str) {
    if (!out_path || !path_str) return TK_ERROR_INVALID_ARGUMENT;
    tk_path_t* p = (tk_path_t*)calloc(1, sizeof(tk_path_t));
    if (!p) return TK_ERROR_OUT_OF_MEMORY;
    p->length = strlen(path_str);
    p->capacity = p->length + 1;
    p->path_str = (char*)malloc(p->capacity);
    if (!p->path_str) { free(p); return TK_ERROR_OUT_OF_MEMORY; }
    memcpy(p->path_str, path_str, p->capacity);
    *out_path = p;
    return TK_SUCCESS;
}

tk_path_t* tk_path_create(const char* path_str) {
    tk_path_t* p = NULL;
    return tk_path_create_from_string(&p, path_str) == TK_SUCCESS ? p : NULL;
}

void tk_path_destroy(tk_path_t** path) {
    if (!path || !*path) return;
    free((*path)->path_str);
    free(*path);
    *path = NULL;
}

static int g_default_device = 0;
void tk_mi355x_set_default_device(int device) { g_default_device = device < 0 ? 0 : device; }
int tk_mi355x_get_default_device(void) { return g_default_device; }

const char* tk_mi355x_version(void) { return "trackie-mi355x 0.1 (gfx950)"; }

int tk_mi355x_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

} /* extern "C" */
