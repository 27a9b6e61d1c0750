/*
 * tk_audio_pipeline.h — the audio pipeline state machine around VAD + ASR and the TTS hand-off:
 *   src/audio/tk_audio_pipeline.h (public API) and tk_audio_pipeline.c: worker thread :550-609, process_chunk ring :387-430,
 *   wake-word phase :480-528, VAD phase :530-548 + process_vad :611-658, process_asr :660-740, VAD events :775-803,
 *   priority TTS queue :837-957, interruption :962-975, process_next_tts_request :977-1010.
 * Same names, argument meaning and error codes.  What is not here, because its engine is a closed or absent third-party component
 * outside the hot path (Porcupine, Piper, the sound classifier): the wake-word DETECTOR, the speech SYNTHESISER and the
 * ambient-sound classifier.  Their hand-off points are kept and made pluggable:
 *   - wake word: with ww_model_path == NULL the pipeline is always awake (every AWAITING_WAKE_WORD phase ends at once); otherwise it waits
 *     for tk_mi355x_audio_pipeline_trigger_wake_word (the call a detector would make);
 *   - TTS: queued requests are handed, in priority order and with the reference's interruption rule, to the synthesiser installed with
 *     tk_mi355x_audio_pipeline_set_synthesizer; without one a request is consumed silently (no audio callback).
 * VAD and ASR are the GPU streams of tk_audio.h.
 */
#ifndef TK_MI355X_AUDIO_PIPELINE_H
#define TK_MI355X_AUDIO_PIPELINE_H

#include "tk_audio.h"
#include "tk_reasoner.h" /* tk_response_priority_e */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tk_audio_pipeline_s tk_audio_pipeline_t;

typedef struct { uint32_t sample_rate; uint32_t channels; } tk_audio_params_t;

typedef enum {
    TK_PIPELINE_STATE_IDLE, TK_PIPELINE_STATE_AWAITING_WAKE_WORD, TK_PIPELINE_STATE_LISTENING_FOR_COMMAND, TK_PIPELINE_STATE_TRANSCRIBING,
    TK_PIPELINE_STATE_SYNTHESIZING
} tk_pipeline_state_e;

typedef struct {
    tk_audio_params_t input_audio_params;
    const char* user_language;
    void* user_data;
    tk_path_t* asr_model_path;
    tk_path_t* vad_model_path;
    tk_path_t* tts_model_path;    /* accepted, unused: no synthesiser is built in */
    tk_path_t* tts_config_path;   /* accepted, unused */
    tk_path_t* ww_model_path;     /* NULL: always awake */
    tk_path_t* ww_keyword_path;   /* accepted, unused */
    float ww_sensitivity;
    tk_path_t* sc_model_path;     /* accepted, unused: the sound classifier is out of scope */
    float vad_silence_threshold_ms;
    float vad_speech_probability_threshold;
} tk_audio_pipeline_config_t;

typedef struct { const char* text; bool is_final; float confidence; } tk_transcription_t;
typedef struct { int sound_class; float confidence; } tk_sound_detection_result_t; /* src/audio/tk_sound_classifier.h:53-59; never produced here */

typedef void (*tk_on_vad_event_cb)(tk_vad_event_e event, void* user_data);
typedef void (*tk_on_transcription_cb)(const tk_transcription_t* result, void* user_data);
typedef void (*tk_on_tts_audio_cb)(const int16_t* audio_data, size_t frame_count, uint32_t sample_rate, void* user_data);
typedef void (*tk_on_tts_interrupt_cb)(void* user_data);
typedef void (*tk_on_ambient_sound_detected_cb)(const tk_sound_detection_result_t* result, void* user_data);

typedef struct {
    tk_on_vad_event_cb on_vad_event;
    tk_on_transcription_cb on_transcription;
    tk_on_tts_audio_cb on_tts_audio_ready;
    tk_on_tts_interrupt_cb on_tts_interrupt;
    tk_on_ambient_sound_detected_cb on_ambient_sound_detected;
} tk_audio_callbacks_t;

TK_API TK_NODISCARD tk_error_code_t tk_audio_pipeline_create(tk_audio_pipeline_t** out_pipeline, const tk_audio_pipeline_config_t* config,
                                                             tk_audio_callbacks_t callbacks);
TK_API void tk_audio_pipeline_destroy(tk_audio_pipeline_t** pipeline);
/* copies the chunk into the 16384-sample ring the worker thread drains; TK_ERROR_BUFFER_TOO_SMALL when it does not fit */
TK_API TK_NODISCARD tk_error_code_t tk_audio_pipeline_process_chunk(tk_audio_pipeline_t* pipeline, const int16_t* audio_chunk, size_t frame_count);
TK_API TK_NODISCARD tk_error_code_t tk_audio_pipeline_synthesize_text(tk_audio_pipeline_t* pipeline, const char* text_to_speak, tk_response_priority_e priority);
TK_API TK_NODISCARD tk_error_code_t tk_audio_pipeline_force_transcription_end(tk_audio_pipeline_t* pipeline);
TK_API tk_pipeline_state_e tk_audio_pipeline_get_state(tk_audio_pipeline_t* pipeline);

/* ---- extensions: the hand-off points of the engines that are out of scope ---- */
/* what a wake-word detector calls on a hit: AWAITING_WAKE_WORD -> LISTENING_FOR_COMMAND (tk_audio_pipeline.c:499-507) */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_audio_pipeline_trigger_wake_word(tk_audio_pipeline_t* pipeline);
/* the synthesiser a TTS request is handed to, on the worker thread: it may call emit(pcm, frames, sample_rate, emit_ctx) any number of
 * times; chunks emitted after a higher-priority request interrupted this one are dropped (tts_audio_callback, :805-825) */
typedef void (*tk_mi355x_tts_emit_fn)(const int16_t* pcm, size_t frame_count, uint32_t sample_rate, void* emit_ctx);
typedef tk_error_code_t (*tk_mi355x_tts_synth_fn)(const char* text, tk_mi355x_tts_emit_fn emit, void* emit_ctx, void* user_data);
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_audio_pipeline_set_synthesizer(tk_audio_pipeline_t* pipeline, tk_mi355x_tts_synth_fn fn, void* user_data);
/* blocks until the ring is drained and the TTS queue is empty (tests and batch hosts; the reference has no such call) */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_audio_pipeline_drain(tk_audio_pipeline_t* pipeline, uint32_t timeout_ms);

#ifdef __cplusplus
}
#endif
#endif
