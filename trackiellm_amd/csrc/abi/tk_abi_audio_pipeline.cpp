/*
 * tk_abi_audio_pipeline.cpp — the audio pipeline state machine (see include/tk/tk_audio_pipeline.h for the reference lines restated).
 * Host code around the GPU VAD / ASR streams: one worker thread, a sample ring, the wake-word -> command -> transcription states,
 * and the priority queue + interruption rule in front of the (pluggable) speech synthesiser.
 */
#include <string.h>
#include <time.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "tk/tk_audio_pipeline.h"

#define TK_AP_RING 16384            /* TK_AUDIO_PIPELINE_INTERNAL_BUFFER_SIZE */
#define TK_AP_VAD_WINDOW_MS 32      /* TK_AUDIO_PIPELINE_VAD_WINDOW_SIZE_MS */
#define TK_AP_MAX_TRANSCRIPTION 1024
#define TK_AP_MAX_TTS 16            /* TK_AUDIO_PIPELINE_MAX_TTS_QUEUE_SIZE */
#define TK_AP_WAKE_FRAME 512        /* Porcupine's frame length at 16 kHz: the unit audio is consumed in while awaiting the wake word */

struct TtsItem { std::string text; tk_response_priority_e priority; bool processing = false; };

struct tk_audio_pipeline_s {
    tk_audio_pipeline_config_t config;
    tk_audio_callbacks_t cb;
    uint32_t sample_rate = 16000, frame_size = 512;
    tk_vad_silero_context_t* vad = nullptr;
    tk_asr_whisper_context_t* asr = nullptr;
    std::atomic<int> state{TK_PIPELINE_STATE_IDLE};
    std::mutex mu; /* the reference's worker_mutex: ring, TTS queue */
    std::mutex asr_mu; /* the ASR segment buffer, the transcription text and the VAD / ASR contexts: the worker holds it across process_vad (mu is
                        * released there), tk_audio_pipeline_force_transcription_end takes it too.  Lock order: mu, then asr_mu. */
    std::condition_variable cv, idle_cv;
    std::vector<int16_t> ring;
    size_t head = 0, tail = 0;
    std::vector<int16_t> asr_buf;
    size_t asr_size = 0;
    bool is_speech_active = false;
    std::string transcription;
    uint64_t transition_ns = 0;
    bool speech_since_transition = false;
    bool always_awake = false;
    std::vector<TtsItem> tts; /* kept sorted: highest priority first, FIFO inside a priority */
    TtsItem* current_tts = nullptr;
    std::atomic<bool> tts_interrupt{false};
    tk_mi355x_tts_synth_fn synth = nullptr;
    void* synth_user = nullptr;
    std::atomic<bool> running{false};
    bool busy = false;
    std::thread worker;
};

static uint64_t wall_ns() {
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ULL + (uint64_t)ts.tv_nsec;
}

static int prio_value(tk_response_priority_e p) { /* get_priority_value (:827-835): higher number = more urgent */
    switch (p) {
        case TK_RESPONSE_PRIORITY_CRITICAL: return 4;
        case TK_RESPONSE_PRIORITY_HIGH: return 3;
        case TK_RESPONSE_PRIORITY_NORMAL: return 2;
        case TK_RESPONSE_PRIORITY_LOW: return 1;
        default: return 0;
    }
}

static size_t ring_fill(const tk_audio_pipeline_s* p) { return (p->head + TK_AP_RING - p->tail) % TK_AP_RING; }

static void reset_asr_state(tk_audio_pipeline_s* p) { p->transcription.clear(); }

/* process_asr (:660-740): final = the accumulated segment; the transcription replaces (final) or extends (partial) the current text.
 * Caller holds asr_mu.  The segment is consumed on EVERY exit — also when the engine fails (a GPU error, a segment longer than the
 * engine's buffer at sample rates above 16 kHz): a failed segment is dropped, never kept to be overrun by the next chunk. */
static tk_error_code_t process_asr(tk_audio_pipeline_s* p, bool is_final) {
    if (p->asr_size == 0) return TK_SUCCESS;
    tk_asr_whisper_result_t* r = nullptr;
    const size_t n = p->asr_size;
    p->asr_size = 0;
    tk_error_code_t rc = tk_asr_whisper_process_audio(p->asr, p->asr_buf.data(), n, is_final, &r);
    if (rc != TK_SUCCESS) {
        if (r) tk_asr_whisper_free_result(&r);
        (void)tk_asr_whisper_reset(p->asr); /* whatever part of the segment the engine buffered goes with it */
        return rc;
    }
    if (r && r->text && r->text[0]) {
        if (is_final) p->transcription.assign(r->text, strnlen(r->text, TK_AP_MAX_TRANSCRIPTION - 1));
        else p->transcription.append(r->text, strnlen(r->text, TK_AP_MAX_TRANSCRIPTION - 1 - p->transcription.size()));
        if (p->cb.on_transcription) {
            tk_transcription_t t{p->transcription.c_str(), is_final, r->confidence};
            p->cb.on_transcription(&t, p->config.user_data);
        }
        if (is_final) reset_asr_state(p);
    }
    if (r) tk_asr_whisper_free_result(&r);
    /* the reference clears the ASR buffer only after a final pass (:735-737): a partial pass at the 30 s limit would leave it full and the
     * following memcpy would overrun; every pass consumes the buffer here (asr_size = 0 above) */
    return TK_SUCCESS;
}

static void vad_event(tk_vad_silero_event_e e, void* u) { /* vad_event_callback (:775-803) */
    tk_audio_pipeline_s* p = (tk_audio_pipeline_s*)u;
    if (p->cb.on_vad_event) p->cb.on_vad_event(e, p->config.user_data);
    if (e == TK_VAD_EVENT_SPEECH_STARTED) {
        p->speech_since_transition = true;
        reset_asr_state(p);
    } else if (p->asr_size > 0) {
        p->state.store(TK_PIPELINE_STATE_TRANSCRIBING);
        (void)process_asr(p, true);
        p->state.store(TK_PIPELINE_STATE_AWAITING_WAKE_WORD); /* "after transcription, go back to waiting for wake word" */
    }
}

/* process_vad (:611-658): events first, then the chunk joins the segment while speech is active */
static void process_vad(tk_audio_pipeline_s* p, const int16_t* chunk, size_t n) {
    std::lock_guard<std::mutex> alk(p->asr_mu); /* the VAD callback (vad_event -> process_asr) runs inside this call, on this thread */
    if (tk_vad_silero_process_audio_with_events(p->vad, chunk, n, vad_event, p) != TK_SUCCESS) return;
    tk_vad_silero_state_t st;
    if (tk_vad_silero_get_state(p->vad, &st) != TK_SUCCESS) return;
    p->is_speech_active = st.is_speech_active;
    if (p->is_speech_active) {
        if (p->asr_size + n > p->asr_buf.size()) (void)process_asr(p, false); /* empties the buffer whether or not the engine succeeded */
        if (p->asr_size + n > p->asr_buf.size()) return;                       /* a chunk longer than the whole 30 s buffer: dropped */
        memcpy(p->asr_buf.data() + p->asr_size, chunk, n * sizeof(int16_t));
        p->asr_size += n;
    }
}

struct EmitCtx { tk_audio_pipeline_s* p; };
static void tts_emit(const int16_t* pcm, size_t n, uint32_t sr, void* ctx) { /* tts_audio_callback (:805-825) */
    tk_audio_pipeline_s* p = ((EmitCtx*)ctx)->p;
    if (p->tts_interrupt.load()) return; /* a higher-priority request interrupted this one: the rest of it is dropped */
    if (p->cb.on_tts_audio_ready) p->cb.on_tts_audio_ready(pcm, n, sr, p->config.user_data);
}

/* is there something the worker can consume right now?  (Waking on "ring not empty" alone would spin with the mutex held whenever the
 * ring holds less than a wake-word frame.) */
static bool has_work(const tk_audio_pipeline_s* p) {
    if (!p->tts.empty() && !p->tts.front().processing) return true;
    switch (p->state.load()) {
        case TK_PIPELINE_STATE_AWAITING_WAKE_WORD: return p->always_awake || ring_fill(p) >= TK_AP_WAKE_FRAME;
        case TK_PIPELINE_STATE_LISTENING_FOR_COMMAND: return ring_fill(p) > 0;
        default: return false;
    }
}

static void worker_main(tk_audio_pipeline_s* p) {
    const uint64_t listen_timeout_ns = 5ULL * 1000000000ULL; /* :552 */
    std::unique_lock<std::mutex> lk(p->mu);
    while (p->running.load()) {
        p->busy = false;
        p->idle_cv.notify_all();
        /* wake at least every 100 ms so the listening timeout fires without new audio (the reference only checks it when audio arrives) */
        p->cv.wait_for(lk, std::chrono::milliseconds(100), [&] { return !p->running.load() || has_work(p); });
        if (!p->running.load()) break;
        p->busy = true;
        switch (p->state.load()) {
            case TK_PIPELINE_STATE_AWAITING_WAKE_WORD:
                if (p->always_awake) { /* no detector configured: the phase ends at once */
                    p->state.store(TK_PIPELINE_STATE_LISTENING_FOR_COMMAND);
                    p->transition_ns = wall_ns();
                    p->speech_since_transition = false;
                } else {
                    while (ring_fill(p) >= TK_AP_WAKE_FRAME) p->tail = (p->tail + TK_AP_WAKE_FRAME) % TK_AP_RING; /* frames go to the detector only */
                }
                break;
            case TK_PIPELINE_STATE_LISTENING_FOR_COMMAND:
                if (!p->always_awake && !p->speech_since_transition && wall_ns() - p->transition_ns > listen_timeout_ns) {
                    p->state.store(TK_PIPELINE_STATE_AWAITING_WAKE_WORD);
                } else if (ring_fill(p) > 0) { /* process_audio_for_vad (:530-548): at most one VAD frame per turn */
                    const size_t n = ring_fill(p) > p->frame_size ? p->frame_size : ring_fill(p);
                    std::vector<int16_t> chunk(n);
                    for (size_t i = 0; i < n; ++i) { chunk[i] = p->ring[p->tail]; p->tail = (p->tail + 1) % TK_AP_RING; }
                    lk.unlock(); /* the reference releases worker_mutex around process_vad */
                    process_vad(p, chunk.data(), n);
                    lk.lock();
                }
                break;
            default: break;
        }
        if (!p->tts.empty() && !p->tts.front().processing) { /* process_next_tts_request (:977-1010) */
            TtsItem item = p->tts.front();
            p->tts.front().processing = true;
            p->current_tts = &p->tts.front();
            p->tts_interrupt.store(false);
            const int before = p->state.exchange(TK_PIPELINE_STATE_SYNTHESIZING);
            tk_mi355x_tts_synth_fn fn = p->synth;
            void* fu = p->synth_user;
            lk.unlock();
            if (fn) { EmitCtx ec{p}; (void)fn(item.text.c_str(), tts_emit, &ec, fu); }
            lk.lock();
            /* the item is removed whatever the synthesiser returned; requests queued meanwhile sit behind or in front of it by priority */
            for (size_t i = 0; i < p->tts.size(); ++i)
                if (p->tts[i].processing) { p->tts.erase(p->tts.begin() + i); break; }
            p->current_tts = nullptr;
            int expect = TK_PIPELINE_STATE_SYNTHESIZING;
            p->state.compare_exchange_strong(expect, before);
        }
    }
    p->busy = false;
    p->idle_cv.notify_all();
}

extern "C" {

tk_error_code_t tk_audio_pipeline_create(tk_audio_pipeline_t** out, const tk_audio_pipeline_config_t* config, tk_audio_callbacks_t callbacks) {
    if (!out || !config) return TK_ERROR_INVALID_ARGUMENT;
    *out = nullptr;
    if (config->input_audio_params.channels != 1) return TK_ERROR_INVALID_ARGUMENT; /* only mono (:166-169) */
    if (!config->asr_model_path || !config->vad_model_path) return TK_ERROR_INVALID_ARGUMENT;
    std::unique_ptr<tk_audio_pipeline_s> p(new tk_audio_pipeline_s());
    p->config = *config;
    p->cb = callbacks;
    p->sample_rate = config->input_audio_params.sample_rate;
    p->frame_size = p->sample_rate * TK_AP_VAD_WINDOW_MS / 1000;
    if (p->frame_size == 0) return TK_ERROR_INVALID_ARGUMENT;
    p->ring.assign(TK_AP_RING, 0);
    p->asr_buf.assign((size_t)p->sample_rate * 30, 0);
    p->always_awake = config->ww_model_path == NULL;
    tk_vad_silero_config_t vc{};
    vc.model_path = config->vad_model_path; vc.sample_rate = p->sample_rate; vc.threshold = config->vad_speech_probability_threshold;
    vc.min_silence_duration_ms = config->vad_silence_threshold_ms;
    tk_error_code_t rc = tk_vad_silero_create(&p->vad, &vc);
    if (rc != TK_SUCCESS) return rc;
    tk_asr_whisper_config_t ac{};
    ac.model_path = config->asr_model_path; ac.language = config->user_language; ac.translate_to_en = false; ac.sample_rate = p->sample_rate;
    ac.user_data = config->user_data; ac.n_threads = 4; ac.max_context = 16384; ac.word_threshold = 0.1f;
    rc = tk_asr_whisper_create(&p->asr, &ac);
    if (rc != TK_SUCCESS) { tk_vad_silero_destroy(&p->vad); return rc; }
    p->state.store(TK_PIPELINE_STATE_AWAITING_WAKE_WORD);
    p->running.store(true);
    p->worker = std::thread(worker_main, p.get());
    *out = p.release();
    return TK_SUCCESS;
}

void tk_audio_pipeline_destroy(tk_audio_pipeline_t** pipeline) {
    if (!pipeline || !*pipeline) return;
    tk_audio_pipeline_s* p = *pipeline;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->running.store(false);
    }
    p->cv.notify_all();
    if (p->worker.joinable()) p->worker.join();
    if (p->asr) tk_asr_whisper_destroy(&p->asr);
    if (p->vad) tk_vad_silero_destroy(&p->vad);
    delete p;
    *pipeline = nullptr;
}

tk_error_code_t tk_audio_pipeline_process_chunk(tk_audio_pipeline_t* p, const int16_t* chunk, size_t n) {
    if (!p || !chunk) return TK_ERROR_INVALID_ARGUMENT;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        /* one slot stays free so that head == tail always means "empty" (the reference's full flag is set and never read: a chunk that
         * fills its ring exactly would be lost) */
        if (n > TK_AP_RING - 1 - ring_fill(p)) return TK_ERROR_BUFFER_TOO_SMALL;
        for (size_t i = 0; i < n; ++i) { p->ring[p->head] = chunk[i]; p->head = (p->head + 1) % TK_AP_RING; }
    }
    p->cv.notify_all();
    return TK_SUCCESS;
}

tk_error_code_t tk_audio_pipeline_synthesize_text(tk_audio_pipeline_t* p, const char* text, tk_response_priority_e priority) {
    if (!p || !text) return TK_ERROR_INVALID_ARGUMENT;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        if (p->tts.size() >= TK_AP_MAX_TTS) return TK_ERROR_BUFFER_TOO_SMALL;
        /* insert before the first queued item of strictly lower priority (:869-895) — never in front of the one being spoken */
        size_t at = p->tts.size();
        for (size_t i = 0; i < p->tts.size(); ++i)
            if (!p->tts[i].processing && prio_value(priority) > prio_value(p->tts[i].priority)) { at = i; break; }
        TtsItem* cur = p->current_tts; /* insertion may move the vector's storage: remember by index */
        size_t cur_idx = cur ? (size_t)(cur - p->tts.data()) : 0;
        p->tts.insert(p->tts.begin() + at, TtsItem{text, priority, false});
        if (cur) { if (at <= cur_idx) ++cur_idx; p->current_tts = &p->tts[cur_idx]; }
        /* interruption (:939-947): a more urgent request stops a LOW / NORMAL one that is being spoken */
        if (p->current_tts && prio_value(priority) > prio_value(p->current_tts->priority) && prio_value(p->current_tts->priority) <= 2) {
            p->tts_interrupt.store(true);
            if (p->cb.on_tts_interrupt) p->cb.on_tts_interrupt(p->config.user_data);
        }
    }
    p->cv.notify_all();
    return TK_SUCCESS;
}

tk_error_code_t tk_audio_pipeline_force_transcription_end(tk_audio_pipeline_t* p) {
    if (!p) return TK_ERROR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(p->mu);
    std::lock_guard<std::mutex> alk(p->asr_mu); /* waits for a process_vad in flight on the worker: one user of the segment and of the ASR context at a time */
    if (!p->is_speech_active) return TK_SUCCESS;
    tk_error_code_t rc = process_asr(p, true);
    p->is_speech_active = false;
    return rc;
}

tk_pipeline_state_e tk_audio_pipeline_get_state(tk_audio_pipeline_t* p) { return p ? (tk_pipeline_state_e)p->state.load() : TK_PIPELINE_STATE_IDLE; }

tk_error_code_t tk_mi355x_audio_pipeline_trigger_wake_word(tk_audio_pipeline_t* p) {
    if (!p) return TK_ERROR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(p->mu);
    if (p->state.load() != TK_PIPELINE_STATE_AWAITING_WAKE_WORD) return TK_ERROR_INVALID_STATE;
    p->state.store(TK_PIPELINE_STATE_LISTENING_FOR_COMMAND);
    p->transition_ns = wall_ns();
    p->speech_since_transition = false;
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_audio_pipeline_set_synthesizer(tk_audio_pipeline_t* p, tk_mi355x_tts_synth_fn fn, void* user_data) {
    if (!p) return TK_ERROR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(p->mu);
    p->synth = fn;
    p->synth_user = user_data;
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_audio_pipeline_drain(tk_audio_pipeline_t* p, uint32_t timeout_ms) {
    if (!p) return TK_ERROR_INVALID_ARGUMENT;
    std::unique_lock<std::mutex> lk(p->mu);
    auto done = [&] {
        if (p->busy || !p->tts.empty()) return false;
        const int st = p->state.load();
        if (st == TK_PIPELINE_STATE_LISTENING_FOR_COMMAND) return p->head == p->tail;
        if (st == TK_PIPELINE_STATE_AWAITING_WAKE_WORD) return p->always_awake ? p->head == p->tail : ring_fill(p) < TK_AP_WAKE_FRAME;
        return false;
    };
    p->cv.notify_all();
    return p->idle_cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), done) ? TK_SUCCESS : TK_ERROR_TIMEOUT;
}

} /* extern "C" */
