/*
 * tk_module_exec.h — executors for the reference's generic plugin path (SURVEY.md §8b "Callers", optional extra).
 *
 * The reference dispatches `tk_module_execute_command(ctx, module, command_name, input, NULL, NULL)` (src/ffi/c_api/tk_ffi_api.c:1115)
 * to whatever executor was registered for the module with `tk_module_register(TkModuleType, ModuleExecutor)`
 * (src/ffi/src/ffi_bridge.rs:1298; executor signature :229-234); today the cortex registers no-op executors for VISION / AUDIO /
 * CORTEX (src/cortex/tk_cortex_main.c:416-448).  This header gives those three modules real ones on the MI355X path:
 *
 *     module               command        input                         does
 *     TK_MODULE_VISION     "detect"       tk_mi355x_cmd_detect_t*       tk_object_detector_detect
 *     TK_MODULE_AUDIO      "transcribe"   tk_mi355x_cmd_transcribe_t*   tk_asr_whisper_process_audio
 *     TK_MODULE_CORTEX     "generate"     tk_mi355x_cmd_generate_t*     tk_llm_runner_prepare_generation + generate_next_token loop
 *
 * TkStatus / TkModuleType values are the reference's (src/ffi/c_api/tk_ffi_api.h:109-137).  The context pointer is the reference's
 * opaque TkContext and is not dereferenced.  Results are owned as by the wrapped entry points (freed with their `*_free_*`).
 */
#ifndef TK_MODULE_EXEC_H
#define TK_MODULE_EXEC_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include "tk_audio.h"
#include "tk_model_runner.h"
#include "tk_vision.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t tk_ffi_status_t; /* the reference's TkStatus */
#define TK_STATUS_OK 0
#define TK_STATUS_ERROR_NULL_POINTER (-1)
#define TK_STATUS_ERROR_INVALID_ARGUMENT (-2)
#define TK_STATUS_ERROR_OPERATION_FAILED (-6)
#define TK_STATUS_ERROR_UNSUPPORTED_FEATURE (-7)

typedef int32_t tk_ffi_module_t; /* the reference's TkModuleType */
#define TK_MODULE_CORTEX 0
#define TK_MODULE_VISION 10
#define TK_MODULE_AUDIO 20

typedef tk_ffi_status_t (*tk_module_executor_t)(void* ctx, tk_ffi_module_t module, const char* command_name, void* input);

typedef struct {
    tk_object_detector_t* detector;
    const tk_video_frame_t* frame;
    tk_detection_result_t* results; /* out: free with tk_object_detector_free_results */
    size_t count;                   /* out */
    tk_error_code_t error;          /* out: the wrapped call's code */
} tk_mi355x_cmd_detect_t;

typedef struct {
    tk_asr_whisper_context_t* asr;
    const int16_t* pcm;
    size_t frame_count;
    bool is_final;
    tk_asr_whisper_result_t* result; /* out (may stay NULL: less than one second buffered and not final): tk_asr_whisper_free_result */
    tk_error_code_t error;
} tk_mi355x_cmd_transcribe_t;

typedef struct {
    tk_llm_runner_t* runner;
    const char* prompt;
    bool use_tool_grammar;
    int32_t max_tokens; /* stop after this many pieces (<= 0: until end of sequence / tool call) */
    char* out_text;     /* caller's buffer: the concatenated pieces, NUL-terminated */
    size_t out_cap;
    size_t out_len;     /* out */
    int32_t n_tokens;   /* out */
    bool tool_call;     /* out: generation ended with the tool-call sentinel */
    tk_error_code_t error;
} tk_mi355x_cmd_generate_t;

/* one executor for the three modules; unknown (module, command) pairs return TK_STATUS_ERROR_UNSUPPORTED_FEATURE */
TK_API tk_ffi_status_t tk_mi355x_module_executor(void* ctx, tk_ffi_module_t module, const char* command_name, void* input);
/* registers it for VISION, AUDIO and CORTEX through the host's own tk_module_register (passed in: the symbol lives in the host) */
TK_API tk_ffi_status_t tk_mi355x_register_modules(tk_ffi_status_t (*host_register)(tk_ffi_module_t, tk_module_executor_t));

#ifdef __cplusplus
}
#endif
#endif
