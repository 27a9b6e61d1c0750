/*
 * tk_cortex.h — the fused perception -> reasoning cycle behind the reference's cortex API
 * (src/cortex/tk_cortex_main.h:44-60 states, :63-77 model paths, :80-88 config, :179-182 callbacks, :211-315 API).
 *
 * Scope: a minimal cortex that wires the three GPU streams and reports state changes —
 *   frame  -> tk_object_detector_detect -> (>= 1 object) -> context string -> LLM      (tk_cortex_main.c:1149-1237, 1323-1379)
 *   PCM    -> VAD (events) -> accumulate -> speech end -> ASR final -> LLM              (tk_cortex_main.c:660-666, 1662-1684)
 * The contextual reasoner (tk_reasoner.h) assembles the prompt and the decision engine's parser reads the response; navigation, sensor
 * fusion and TTS synthesis stay outside (§8 "out of scope"): on_tts_audio_ready is never invoked and tk_cortex_inject_sensor_event
 * accepts and drops the event.
 * model paths accept the synthetic:// forms of the per-stream headers; a NULL path picks that stream's synthetic default — unless ALL of
 * llm / object_detection / depth / asr / vad are NULL: that cortex has no engines, only its contextual reasoner (what the reference's
 * tests/tk_cortex_full_test.c:20-34 creates: "All paths are NULL", gpu_device_id -1); it runs without a GPU and refuses injected
 * frames / audio with TK_ERROR_INVALID_STATE.
 * Unlike the reference (which stores the caller's frame pointer, tk_cortex_main.c:682), injected frames are copied.
 */
#ifndef TK_MI355X_CORTEX_H
#define TK_MI355X_CORTEX_H

#include "tk_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tk_cortex_s tk_cortex_t;

typedef enum {
    TK_STATE_UNINITIALIZED,
    TK_STATE_INITIALIZING,
    TK_STATE_IDLE,
    TK_STATE_LISTENING,
    TK_STATE_PROCESSING,
    TK_STATE_RESPONDING,
    TK_STATE_SHUTDOWN,
    TK_STATE_FATAL_ERROR
} tk_system_state_e;

typedef struct {
    const char* llm_model;
    const char* object_detection_model;
    const char* depth_estimation_model; /* NULL / "": no depth analysis; an ONNX file of the convolutional MiDaS class (tk_depth.h) */
    const char* asr_model;
    const char* tts_model_dir;          /* ignored */
    const char* vad_model;
    const char* tesseract_data_dir;     /* ignored */
} tk_model_paths_t;

typedef struct {
    tk_model_paths_t model_paths;
    int gpu_device_id; /* -1 in the reference means CPU; here it selects device 0 (no CPU path) */
    float main_loop_frequency_hz;
    const char* user_language;
    void* user_data;
} tk_cortex_config_t;

typedef struct tk_sensor_event_s tk_sensor_event_t; /* sensor fusion is out of scope: opaque */

typedef void (*tk_on_state_change_cb)(tk_system_state_e new_state, void* user_data);
typedef void (*tk_on_tts_audio_ready_cb)(const int16_t* audio_data, size_t frame_count, uint32_t sample_rate, void* user_data);

typedef struct {
    tk_on_state_change_cb on_state_change;
    tk_on_tts_audio_ready_cb on_tts_audio_ready;
} tk_cortex_callbacks_t;

TK_API TK_NODISCARD tk_error_code_t tk_cortex_create(tk_cortex_t** out_cortex, const tk_cortex_config_t* config, tk_cortex_callbacks_t callbacks);
TK_API void tk_cortex_destroy(tk_cortex_t** cortex);
TK_API TK_NODISCARD tk_error_code_t tk_cortex_run(tk_cortex_t* cortex);  /* blocks until tk_cortex_stop */
TK_API TK_NODISCARD tk_error_code_t tk_cortex_stop(tk_cortex_t* cortex);
TK_API TK_NODISCARD tk_error_code_t tk_cortex_inject_audio_frame(tk_cortex_t* cortex, const int16_t* audio_data, size_t frame_count);
TK_API TK_NODISCARD tk_error_code_t tk_cortex_inject_video_frame(tk_cortex_t* cortex, const tk_video_frame_t* frame);
TK_API TK_NODISCARD tk_error_code_t tk_cortex_inject_sensor_event(tk_cortex_t* cortex, const tk_sensor_event_t* event);
TK_API TK_NODISCARD tk_error_code_t tk_cortex_get_state(const tk_cortex_t* cortex, tk_system_state_e* out_state);
/* src/cortex/tk_cortex_main.h:321-331 ("for testing purposes only"): the cortex's contextual reasoner, or NULL */
struct tk_contextual_reasoner_s;
TK_API struct tk_contextual_reasoner_s* tk_cortex_get_contextual_reasoner(tk_cortex_t* cortex);

/* ---- extensions ---- */
typedef struct {
    uint64_t frames_processed, frames_with_objects, speech_segments, llm_responses, llm_tokens, events_dropped;
    uint64_t responses_parsed, actions_parsed; /* LLM responses that were the decision engine's JSON schema, and the actions they carried */
} tk_mi355x_cortex_stats_t;
TK_API void tk_mi355x_cortex_get_stats(const tk_cortex_t* cortex, tk_mi355x_cortex_stats_t* out);
/* text of the most recent LLM response (copied into buf, NUL terminated); returns its length */
TK_API size_t tk_mi355x_cortex_last_response(const tk_cortex_t* cortex, char* buf, size_t cap);
/* the prompt of the most recent LLM turn: the contextual reasoner's context string (src/cortex/tk_contextual_reasoner.c:681-743) */
TK_API size_t tk_mi355x_cortex_last_prompt(const tk_cortex_t* cortex, char* buf, size_t cap);
TK_API void tk_mi355x_cortex_set_max_response_tokens(tk_cortex_t* cortex, int n);

#ifdef __cplusplus
}
#endif
#endif
