/* TEST INFRASTRUCTURE: trivial stand-ins for the GPU VAD / ASR entry points (amplitude threshold instead of a network, "seg<N>" instead of a
 * transcription), so that the audio pipeline's HOST logic — ring, worker thread, wake-word / listening / transcribing states, the TTS priority
 * queue and its interruption rule (csrc/abi/tk_abi_audio_pipeline.cpp) — runs on a machine without a GPU.  Linked with that one product
 * source file into a throw-away .so by tests/test_audio_pipeline_cpu.py; never part of the product library. */
#include "tk/tk_audio_pipeline.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
struct tk_vad_silero_context_s { int n; bool active; };
struct tk_asr_whisper_context_s { int n; };
extern "C" {
void tk_error_set_detail(const char*, ...) {}
tk_error_code_t tk_vad_silero_create(tk_vad_silero_context_t** o, const tk_vad_silero_config_t*) { *o = new tk_vad_silero_context_s{0,false}; return TK_SUCCESS; }
void tk_vad_silero_destroy(tk_vad_silero_context_t** c) { delete *c; *c = nullptr; }
tk_error_code_t tk_vad_silero_process_audio_with_events(tk_vad_silero_context_t* c, const int16_t* a, size_t n, tk_vad_silero_event_callback_t cb, void* u) {
    usleep(800); c->n += (int)n; bool loud = false; for (size_t i = 0; i < n; ++i) loud |= abs(a[i]) > 1000;
    if (loud && !c->active) { c->active = true; cb(TK_VAD_EVENT_SPEECH_STARTED, u); }
    if (!loud && c->active) { c->active = false; cb(TK_VAD_EVENT_SPEECH_ENDED, u); }
    return TK_SUCCESS; }
tk_error_code_t tk_vad_silero_get_state(tk_vad_silero_context_t* c, tk_vad_silero_state_t* s) { memset(s, 0, sizeof *s); s->is_speech_active = c->active; return TK_SUCCESS; }
tk_error_code_t tk_asr_whisper_create(tk_asr_whisper_context_t** o, const tk_asr_whisper_config_t*) { *o = new tk_asr_whisper_context_s{0}; return TK_SUCCESS; }
void tk_asr_whisper_destroy(tk_asr_whisper_context_t** c) { delete *c; *c = nullptr; }
static int g_asr_fail = 0, g_asr_calls = 0, g_asr_resets = 0; static size_t g_asr_max_n = 0;
__attribute__((visibility("default"))) void stub_asr_set_fail(int on) { g_asr_fail = on; }
__attribute__((visibility("default"))) void stub_asr_stats(int* calls, int* resets, size_t* max_n) { *calls = g_asr_calls; *resets = g_asr_resets; *max_n = g_asr_max_n; }
tk_error_code_t tk_asr_whisper_reset(tk_asr_whisper_context_t*) { ++g_asr_resets; return TK_SUCCESS; }
tk_error_code_t tk_asr_whisper_process_audio(tk_asr_whisper_context_t*, const int16_t*, size_t n, bool fin, tk_asr_whisper_result_t** out) {
    ++g_asr_calls; if (n > g_asr_max_n) g_asr_max_n = n;
    if (g_asr_fail) { *out = nullptr; return TK_ERROR_BUFFER_TOO_SMALL; }
    tk_asr_whisper_result_t* r = (tk_asr_whisper_result_t*)calloc(1, sizeof *r); char b[64]; snprintf(b, 64, "seg%zu", n); r->text = strdup(b); r->text_length = strlen(b); r->confidence = 0.9f; r->is_partial = !fin; *out = r; return TK_SUCCESS; }
void tk_asr_whisper_free_result(tk_asr_whisper_result_t** r) { free((*r)->text); free(*r); *r = nullptr; }
}
extern "C" {
__attribute__((visibility("default"))) tk_path_t* tk_path_create(const char* s) { tk_path_t* p = (tk_path_t*)calloc(1, sizeof *p); p->path_str = strdup(s); return p; }
__attribute__((visibility("default"))) void tk_path_destroy(tk_path_t** p) { free((*p)->path_str); free(*p); *p = nullptr; }
__attribute__((visibility("default"))) const char* tk_error_get_detail() { return ""; }
__attribute__((visibility("default"))) const char* tk_error_to_string(tk_error_code_t) { return ""; }
__attribute__((visibility("default"))) const char* tk_mi355x_version() { return "stub"; }
}
