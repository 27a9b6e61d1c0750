/*
 * tk_types.h — plain data types that cross the tk_* boundary.
 *   tk_path_t          : the hot-path sources dereference `->path_str`
 *                        (src/vision/tk_object_detector.c:104, src/audio/tk_asr_whisper.c:237,
 *                        src/ai_models/tk_model_loader.c:251) although the reference struct is
 *                        {buffer,length,capacity} (src/internal_tools/tk_file_manager.c:60-64);
 *                        first member named path_str, same 3-field size.
 *   tk_video_frame_t   : src/cortex/tk_cortex_main.h:98-104 (the C header wins over the Rust mirrors)
 *   tk_rect_t          : src/vision/tk_vision_pipeline.h:158-163
 *   tk_vision_backend_e: src/vision/tk_vision_pipeline.h:56-61
 */
#ifndef TK_MI355X_TYPES_H
#define TK_MI355X_TYPES_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include "tk_error_handling.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tk_path_s {
    char* path_str;
    size_t length;
    size_t capacity;
} tk_path_t;

/* src/internal_tools/tk_file_manager.h:119,172; tests/tk_asr_whisper_test.c:52 calls tk_path_create(const char*) */
TK_API TK_NODISCARD tk_error_code_t tk_path_create_from_string(tk_path_t** out_path, const char* path_str);
TK_API tk_path_t* tk_path_create(const char* path_str);
TK_API void tk_path_destroy(tk_path_t** path);

typedef enum { TK_PIXEL_FORMAT_RGB8, TK_PIXEL_FORMAT_RGBA8 } tk_pixel_format_e;

typedef struct tk_video_frame_s {
    uint32_t width;
    uint32_t height;
    uint32_t stride; /* bytes per row; honoured here (the reference CPU preprocessor assumes width*3) */
    tk_pixel_format_e format;
    const uint8_t* data;
} tk_video_frame_t;

typedef struct { int x, y, w, h; } tk_rect_t;

typedef enum {
    TK_VISION_BACKEND_CPU,
    TK_VISION_BACKEND_CUDA,
    TK_VISION_BACKEND_METAL,
    TK_VISION_BACKEND_ROCM
} tk_vision_backend_e;

#ifdef __cplusplus
}
#endif
#endif
