/*
 * tk_model_runner.h — model loader + LLM runner surface of the tk_* C-ABI.
 *
 * Loader: src/ai_models/tk_model_loader.h:381-421 with the slim parameter structs the only
 *         real callers construct (src/ai_models/src/lib.rs:127-147; SURVEY.md Appendix B —
 *         the C header's >100-field struct has a duplicate member and does not compile).
 * Runner: src/ai_models/tk_model_runner.h:135-221; `tk_llm_runner_create` takes the loader's
 *         void* handle as the definition does (src/ai_models/tk_runner_lifecycle.c:17-27).
 * Behaviour restated from src/ai_models/tk_runner_streaming.c:13-85 and
 * tk_runner_helpers.c:78-138; sampling is argmax.
 *
 * model_path forms accepted by tk_model_loader_load_model:
 *   "/path/model.gguf"                         GGUF v3, llama arch, F32/Q4_K/Q6_K tensors
 *   "synthetic://mistral-7b?seed=4"            Mistral-7B-v0.1-shaped, Q4_K_M recipe, seeded weights
 *   "synthetic://tiny?seed=4"                  2-layer test geometry
 */
#ifndef TK_MI355X_MODEL_RUNNER_H
#define TK_MI355X_MODEL_RUNNER_H

#include "tk_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tk_model_loader_s tk_model_loader_t;

typedef enum { TK_MODEL_FORMAT_UNKNOWN = 0, TK_MODEL_FORMAT_GGUF, TK_MODEL_FORMAT_ONNX } tk_model_format_e;

typedef struct {
    uint32_t max_models;
    uint32_t num_threads;
} tk_model_loader_config_t;

typedef struct {
    tk_path_t* model_path;
    uint32_t model_type; /* tk_model_format_e */
    bool force_reload;
    uint32_t gpu_layers; /* ignored: all layers are resident in HBM */
    uint32_t cpu_threads;
    bool use_mmap;
    bool use_mlock;
    bool numa;
    uint32_t seed;
    const char* lora_adapter; /* NULL / "" = none; else a "ggla" or GGUF adapter merged into the weights at load (tk_model_loader.c:259-270) */
} tk_model_load_params_t;

TK_API TK_NODISCARD tk_error_code_t tk_model_loader_create(tk_model_loader_t** out_loader, const tk_model_loader_config_t* config);
TK_API void tk_model_loader_destroy(tk_model_loader_t** loader);
TK_API TK_NODISCARD tk_error_code_t tk_model_loader_load_model(tk_model_loader_t* loader, const tk_model_load_params_t* params,
                                                               void** out_model_handle);
TK_API TK_NODISCARD tk_error_code_t tk_model_loader_unload_model(tk_model_loader_t* loader, void** model_handle);

typedef struct tk_llm_runner_s tk_llm_runner_t;
typedef struct tk_llm_result_s tk_llm_result_t;

typedef struct {
    uint32_t context_size;
    const char* system_prompt;
    uint32_t random_seed;
} tk_llm_config_t;

typedef enum { TK_LLM_RESULT_TYPE_UNKNOWN, TK_LLM_RESULT_TYPE_TEXT_RESPONSE, TK_LLM_RESULT_TYPE_TOOL_CALL } tk_llm_result_type_e;

typedef struct {
    char* name;
    char* arguments_json;
} tk_llm_tool_call_t;

struct tk_llm_result_s {
    tk_llm_result_type_e type;
    union {
        char* text_response;
        tk_llm_tool_call_t tool_call;
    } data;
};

/* sentinel returned by generate_next_token when a tool call completed (tk_runner_streaming.c:55) */
#define TK_TOOL_CALL_TOKEN ((const char*)1)

TK_API TK_NODISCARD tk_error_code_t tk_llm_runner_create(tk_llm_runner_t** out_runner, void* model_handle, const tk_llm_config_t* config);
TK_API void tk_llm_runner_destroy(tk_llm_runner_t** runner);
TK_API TK_NODISCARD tk_error_code_t tk_llm_runner_prepare_generation(tk_llm_runner_t* runner, const char* prompt, bool use_tool_grammar);
TK_API TK_NODISCARD const char* tk_llm_runner_generate_next_token(tk_llm_runner_t* runner);
TK_API TK_NODISCARD tk_error_code_t tk_llm_runner_add_tool_response(tk_llm_runner_t* runner, const char* tool_name, const char* tool_output);
TK_API TK_NODISCARD tk_error_code_t tk_llm_runner_reset_context(tk_llm_runner_t* runner);
TK_API void tk_llm_result_destroy(tk_llm_result_t** result);

#ifdef __cplusplus
}
#endif
#endif
