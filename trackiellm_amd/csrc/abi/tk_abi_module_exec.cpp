/* tk_abi_module_exec.cpp — executors for the reference's tk_module_register plugin path (include/tk/tk_module_exec.h). */
#include <string.h>

#include "tk/tk_module_exec.h"

extern "C" {

tk_ffi_status_t tk_mi355x_module_executor(void* /*ctx*/, tk_ffi_module_t module, const char* command_name, void* input) {
    if (!command_name || !input) return TK_STATUS_ERROR_NULL_POINTER;
    if (module == TK_MODULE_VISION && strcmp(command_name, "detect") == 0) {
        auto* c = (tk_mi355x_cmd_detect_t*)input;
        if (!c->detector || !c->frame) return TK_STATUS_ERROR_NULL_POINTER;
        c->results = nullptr;
        c->count = 0;
        c->error = tk_object_detector_detect(c->detector, c->frame, &c->results, &c->count);
        return c->error == TK_SUCCESS ? TK_STATUS_OK : TK_STATUS_ERROR_OPERATION_FAILED;
    }
    if (module == TK_MODULE_AUDIO && strcmp(command_name, "transcribe") == 0) {
        auto* c = (tk_mi355x_cmd_transcribe_t*)input;
        if (!c->asr || (!c->pcm && c->frame_count)) return TK_STATUS_ERROR_NULL_POINTER;
        c->result = nullptr;
        c->error = tk_asr_whisper_process_audio(c->asr, c->pcm, c->frame_count, c->is_final, &c->result);
        return c->error == TK_SUCCESS ? TK_STATUS_OK : TK_STATUS_ERROR_OPERATION_FAILED;
    }
    if (module == TK_MODULE_CORTEX && strcmp(command_name, "generate") == 0) {
        auto* c = (tk_mi355x_cmd_generate_t*)input;
        if (!c->runner || !c->prompt || !c->out_text) return TK_STATUS_ERROR_NULL_POINTER;
        if (c->out_cap == 0) return TK_STATUS_ERROR_INVALID_ARGUMENT;
        c->out_len = 0; c->n_tokens = 0; c->tool_call = false; c->out_text[0] = 0;
        c->error = tk_llm_runner_prepare_generation(c->runner, c->prompt, c->use_tool_grammar);
        if (c->error != TK_SUCCESS) return TK_STATUS_ERROR_OPERATION_FAILED;
        for (;;) {
            if (c->max_tokens > 0 && c->n_tokens >= c->max_tokens) break;
            const char* piece = tk_llm_runner_generate_next_token(c->runner);
            if (!piece) break;                                            /* end of sequence (or an engine error: the runner keeps the detail) */
            if (piece == TK_TOOL_CALL_TOKEN) { c->tool_call = true; break; } /* src/ai_models/tk_runner_streaming.c:55 */
            ++c->n_tokens;
            const size_t n = strlen(piece);
            if (c->out_len + n + 1 > c->out_cap) { c->error = TK_ERROR_BUFFER_TOO_SMALL; return TK_STATUS_ERROR_INVALID_ARGUMENT; }
            memcpy(c->out_text + c->out_len, piece, n);
            c->out_len += n;
            c->out_text[c->out_len] = 0;
        }
        return TK_STATUS_OK;
    }
    return TK_STATUS_ERROR_UNSUPPORTED_FEATURE;
}

tk_ffi_status_t tk_mi355x_register_modules(tk_ffi_status_t (*host_register)(tk_ffi_module_t, tk_module_executor_t)) {
    if (!host_register) return TK_STATUS_ERROR_NULL_POINTER;
    for (tk_ffi_module_t m : {TK_MODULE_VISION, TK_MODULE_AUDIO, TK_MODULE_CORTEX}) {
        const tk_ffi_status_t st = host_register(m, tk_mi355x_module_executor);
        if (st != TK_STATUS_OK) return st;
    }
    return TK_STATUS_OK;
}

} /* extern "C" */
