/*
 * tk_audio.h — ASR + VAD streams of the tk_* C-ABI.
 *   tk_asr_whisper_*   src/audio/tk_asr_whisper.h:36-47 (config), :53-58 (result), :82-162 (API);
 *                      behaviour of src/audio/tk_asr_whisper.c:282-344 (30 s int16 buffer, overflow => reset,
 *                      < 16000 samples and !final => empty result, final => buffer cleared, confidence = 0.9 constant)
 *   tk_vad_silero_*    src/sensors/tk_vad_silero.h (API), src/sensors/tk_vad_silero.c:283-322 (state machine),
 *                      :327-390 (30 ms window / 10 ms hop, time advances by the WINDOW length per step — kept),
 *                      :393-470 (defaults), :488-600
 * model_path forms: "synthetic://whisper-tiny.en?seed=6" | a TKWHSP1 container;  "synthetic://vad?seed=7".
 * The single event enum is shared by both reference headers that define it.
 */
#ifndef TK_MI355X_AUDIO_H
#define TK_MI355X_AUDIO_H

#include "tk_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tk_asr_whisper_context_s tk_asr_whisper_context_t;

typedef struct {
    tk_path_t* model_path;
    const char* language;
    bool translate_to_en;
    uint32_t sample_rate; /* must be 16000 */
    void* user_data;
    int n_threads;        /* ignored */
    int max_context;
    float word_threshold;
} tk_asr_whisper_config_t;

typedef struct {
    char* text;
    size_t text_length;
    float confidence;
    bool is_partial;
} tk_asr_whisper_result_t;

TK_API TK_NODISCARD tk_error_code_t tk_asr_whisper_create(tk_asr_whisper_context_t** out_context, const tk_asr_whisper_config_t* config);
TK_API void tk_asr_whisper_destroy(tk_asr_whisper_context_t** context);
TK_API TK_NODISCARD tk_error_code_t tk_asr_whisper_process_audio(tk_asr_whisper_context_t* context, const int16_t* audio_data, size_t frame_count,
                                                                 bool is_final, tk_asr_whisper_result_t** out_result);
TK_API void tk_asr_whisper_free_result(tk_asr_whisper_result_t** result);
TK_API TK_NODISCARD tk_error_code_t tk_asr_whisper_reset(tk_asr_whisper_context_t* context);
TK_API TK_NODISCARD tk_error_code_t tk_asr_whisper_set_language(tk_asr_whisper_context_t* context, const char* language);

typedef struct tk_vad_silero_context_s tk_vad_silero_context_t;

typedef struct {
    tk_path_t* model_path;
    uint32_t sample_rate; /* 8000, 16000 or 48000 */
    void* user_data;
    float threshold;
    float min_silence_duration_ms;
    float min_speech_duration_ms;
    float speech_pad_ms;
} tk_vad_silero_config_t;

typedef struct {
    bool is_speech_active;
    float speech_probability;
    float silence_duration_ms;
    float speech_duration_ms;
} tk_vad_silero_state_t;

typedef enum { TK_VAD_EVENT_SPEECH_STARTED, TK_VAD_EVENT_SPEECH_ENDED } tk_vad_event_e;
typedef tk_vad_event_e tk_vad_silero_event_e;
typedef void (*tk_vad_silero_event_callback_t)(tk_vad_silero_event_e event, void* user_data);

TK_API TK_NODISCARD tk_error_code_t tk_vad_silero_create(tk_vad_silero_context_t** out_context, const tk_vad_silero_config_t* config);
TK_API void tk_vad_silero_destroy(tk_vad_silero_context_t** context);
TK_API TK_NODISCARD tk_error_code_t tk_vad_silero_process_audio(tk_vad_silero_context_t* context, const int16_t* audio_data, size_t frame_count,
                                                                float* out_probability);
TK_API TK_NODISCARD tk_error_code_t tk_vad_silero_process_audio_with_events(tk_vad_silero_context_t* context, const int16_t* audio_data,
                                                                            size_t frame_count, tk_vad_silero_event_callback_t callback,
                                                                            void* user_data);
TK_API TK_NODISCARD tk_error_code_t tk_vad_silero_get_state(tk_vad_silero_context_t* context, tk_vad_silero_state_t* out_state);
TK_API TK_NODISCARD tk_error_code_t tk_vad_silero_reset(tk_vad_silero_context_t* context);
TK_API TK_NODISCARD tk_error_code_t tk_vad_silero_set_threshold(tk_vad_silero_context_t* context, float threshold);

/* ---- extensions (no reference counterpart) ---- */
typedef struct {
    int32_t n_mels, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
    int32_t n_text_ctx, n_text_state, n_text_head, n_text_layer, n_vocab;
} tk_mi355x_whisper_hparams_t;
/* explicit geometry + batch (parity tests use a small geometry; bench uses tiny.en with max_batch = B); max_batch in [1, 256] */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_asr_create(tk_asr_whisper_context_t** out, const tk_mi355x_whisper_hparams_t* hp, uint64_t seed,
                                                         int device, int max_batch);
/* B utterances of n_samples int16 each -> exactly n_steps greedy tokens per utterance (forced, EOT ignored);
 * optional outputs: mel [B][2*ctx][n_mels], enc [B][ctx][d], logits of the first sampled position [B][n_vocab] */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_asr_transcribe_tokens(tk_asr_whisper_context_t* ctx, int batch, const int16_t* pcm, int n_samples,
                                                                    int n_steps, int32_t* tokens_out, float* mel_out, float* enc_out,
                                                                    float* logits_out);
TK_API void tk_mi355x_asr_set_decode_steps(tk_asr_whisper_context_t* ctx, int n_steps);
/* tk_asr_whisper_process_audio decodes the way whisper_full does under the reference's parameters (src/audio/tk_asr_whisper.c:89-110:
 * suppress_blank off, suppress_non_speech_tokens on, timestamps on): whisper.cpp's whisper_process_logits and the token bookkeeping of its decode
 * loop run on the device, the text is the text tokens of the sequence up to its result length, at most `decode_steps` tokens per call.
 * enable = 0 restores the earlier rounds' decode (<|notimestamps|> prompt, bare arg max, exactly decode_steps tokens); vocabularies without
 * whisper's special-token layout (fewer than 51864 tokens) always decode that way. */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_asr_set_reference_decode(tk_asr_whisper_context_t* ctx, int enable);
/* Opt-in fast contraction (the detector's twin is tk_mi355x_detector_set_fast_contraction): the long passes of a transcription
 * — log-mel, the encoder's convolutions and linear layers, and its attention as one fused kernel (no score matrix in memory) — contract on the f16 matrix pipe with every operand split into two f16 halves
 * (~22 significant bits, fp32 accumulation): within ~1e-6 of the exact chains' scale, not their bits.  Decoder steps keep the exact path.  Off by
 * default; the exact path stays the parity path and the checker (tests/test_audio_gpu.py::test_asr_fast_contraction_gate). */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_asr_set_fast_contraction(tk_asr_whisper_context_t* ctx, int on);
/* one utterance through that decode: tokens [n_steps] (eot behind the row's end), log-probabilities [n_steps] (may be NULL), *result_len = tokens that
 * make up the text, *status = 0 (n_steps reached), 1 (completed), 2 (failed: whisper.cpp would fall back to the next temperature) */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_asr_transcribe_ref(tk_asr_whisper_context_t* ctx, const int16_t* pcm, int n_samples, int n_steps, float temperature,
                                                                 uint64_t seed, int32_t* tokens_out, float* logprobs_out, int32_t* result_len, int32_t* status);
/* the static suppression table of that decode ([n_vocab], 1 = never sampled) and the ids of the first timestamp / end-of-text tokens; returns the
 * table's length (0: the vocabulary has no timestamp tokens) */
TK_API int32_t tk_mi355x_asr_suppress_table(tk_asr_whisper_context_t* ctx, uint8_t* out, int32_t cap, int32_t* token_beg, int32_t* token_eot);
/* the per-model-file registry behind tk_asr_whisper_create: contexts opened on the same checkpoint / device share the weights and one batched
 * engine whose scheduler coalesces their one-utterance calls (tk_asr_whisper_process_audio).  Counters of this context's shared engine: live
 * contexts, batched jobs run, utterances they carried, the widest job.  Any pointer may be NULL. */
TK_API void tk_mi355x_asr_share_stats(const tk_asr_whisper_context_t* ctx, uint64_t* handles, uint64_t* batches, uint64_t* utterances, uint64_t* widest);
/* whisper.cpp's decoding policy, which the reference's wrapper arms with temperature_inc 0.2 / entropy_thold 2.4 / logprob_thold -1.0
 * (src/audio/tk_asr_whisper.c:126-138; partial results: no fallback, :137).  OFF by default: the plain greedy decode is what the parity suite pins.
 *   transcribe_policy: the forced decode of transcribe_tokens with the token of every step picked by temperature (0 = arg max; > 0 = one draw from
 *     softmax(l / temperature) restricted to the 64 largest logits, by the library's counter-based sampler keyed (seed, position x batch + b)) and,
 *     optionally, logprobs_out [batch][n_steps] = log-probability of each produced token under softmax(l / temperature) over the whole vocabulary;
 *   set_decode_policy: tk_asr_whisper_process_audio then decodes a FINAL result at temperatures 0, inc, 2 inc, ... <= 1 until the decode passes
 *     whisper.cpp's test (mean log-probability of the tokens up to and including end-of-text >= logprob_thold, and — past 32 tokens — entropy of
 *     the histogram of the last 32 tokens >= entropy_thold); attempt a uses seed + a; a partial result decodes once at temperature 0;
 *   last_decode: temperature, mean log-probability and attempt count of the decode the last process_audio returned.
 * TK_MI355X_ASR_POLICY=1 in the environment arms the policy with the reference's three numbers in tk_asr_whisper_create (for hosts that only
 * know the reference's entry points).  Not restated from whisper.cpp: its logit filters (blank / non-speech / timestamp rules), the no-speech
 * probability and the best-of-5 decoders it runs at temperature > 0 — one decoder per temperature here. */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_asr_transcribe_policy(tk_asr_whisper_context_t* ctx, int batch, const int16_t* pcm, int n_samples,
                                                                    int n_steps, float temperature, uint64_t seed, int32_t* tokens_out,
                                                                    float* logprobs_out);
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_asr_set_decode_policy(tk_asr_whisper_context_t* ctx, int enable, float temperature_inc,
                                                                    float entropy_thold, float logprob_thold, uint64_t seed);
TK_API void tk_mi355x_asr_last_decode(const tk_asr_whisper_context_t* ctx, float* temperature, float* avg_logprob, int32_t* attempts);
/* the decoder prompt the next transcription starts from (English-only: <|sot|><|notimestamps|>; multilingual: + language and task
 * tokens, whisper.cpp's whisper_full order); returns the count, -1 when the configured language has no token */
TK_API int tk_mi355x_asr_prompt_tokens(tk_asr_whisper_context_t* ctx, int32_t* out, int cap);
/* geometry of a created context (a whisper.cpp ggml checkpoint brings its own) */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_asr_get_hparams(tk_asr_whisper_context_t* ctx, tk_mi355x_whisper_hparams_t* out);
/* whisper.cpp "ggml" checkpoint (the file tk_asr_whisper_config_t.model_path names, src/audio/tk_asr_whisper.c:238): parse header,
 * filter bank, vocabulary and tensor directory and check that every tensor of the graph is present; no GPU involved */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_whisper_ggml_probe(const char* path, tk_mi355x_whisper_hparams_t* out, int32_t* n_tokens,
                                                                 int32_t* n_tensors);
/* feed one window probability straight into the VAD state machine (30 ms step); returns -1 none, 0 started, 1 ended */
/* a VAD .onnx file (tk_vad_silero_config_t.model_path, src/sensors/tk_vad_silero.c:110-150): parse the graph (no ONNX Runtime, no GPU) and
 * check that every node is an op the GPU executor runs (csrc/audio/tk_vad_graph.h); counts optional */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_vad_onnx_probe(const char* path, int32_t* n_nodes, int32_t* n_initialisers, int32_t* n_state_inputs);
TK_API int tk_mi355x_vad_step(tk_vad_silero_context_t* ctx, float probability);
/* probabilities of n consecutive float windows of the model's window length */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_vad_probabilities(tk_vad_silero_context_t* ctx, const float* windows, int n, float* out);

#ifdef __cplusplus
}
#endif
#endif
