/*
 * tk_abi_cortex.cpp — minimal cortex: event queue (128, src/cortex/tk_cortex_main.c:527), one loop thread
 * (:957-994), video ring of 4 (:542), VAD -> ASR -> LLM and detect -> LLM triggers (:1081, :1224-1237).
 *
 * The reference runs detect -> prompt -> whole LLM response serially on the loop thread (:1149-1237, :1323-1379).  Here the three
 * model streams run concurrently, each on its own host thread and HIP stream: VAD + ASR on the thread that injects audio (the
 * reference's audio worker, src/audio/tk_audio_pipeline.c:550-609), the detector on the loop thread (tk_cortex_run), the LLM on a
 * response thread fed by a small prompt queue — so the detector already works on frame k + 1 while the LLM answers frame k, and the
 * LLM rows of K cortex handles that share a model file are decoded together (csrc/llm/tk_llm_batcher.h).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "tk/tk_audio.h"
#include "tk/tk_cortex.h"
#include "tk/tk_model_runner.h"
#include "tk/tk_reasoner.h"
#include "tk/tk_vision.h"

#define TK_CORTEX_QUEUE_CAP 128
#define TK_CORTEX_VIDEO_RING 4
#define TK_CORTEX_PROMPT_QUEUE 8 /* prompts waiting for the response thread; the oldest scene prompt is dropped when full */

struct CortexEvent {
    enum { VIDEO, SPEECH } kind;
    int slot = 0;            /* video ring slot */
    std::string text;        /* final transcription */
};

struct tk_cortex_s {
    tk_cortex_config_t config;
    tk_cortex_callbacks_t cb;
    std::atomic<int> state{TK_STATE_UNINITIALIZED};
    std::atomic<bool> stop{false};
    std::mutex mu;
    std::condition_variable cv;
    std::deque<CortexEvent> queue;
    struct Frame { tk_video_frame_t f; std::vector<uint8_t> data; } ring[TK_CORTEX_VIDEO_RING];
    int ring_next = 0;
    tk_vision_pipeline_t* vis = nullptr; /* detection + depth + fusion, as the reference's cortex owns a vision pipeline (tk_cortex_main.c:773-790) */
    tk_asr_whisper_context_t* asr = nullptr;
    tk_vad_silero_context_t* vad = nullptr;
    tk_model_loader_t* loader = nullptr;
    void* llm_model = nullptr;
    tk_llm_runner_t* runner = nullptr;
    tk_contextual_reasoner_t* reasoner = nullptr; /* scene / conversation memory the prompt is assembled from (tk_cortex_main.c:1323-1340) */
    std::mutex audio_mu;     /* VAD + speech accumulation happen on the injecting thread, like the reference's audio worker */
    std::vector<int16_t> speech;
    bool in_speech = false;
    int max_tokens = 32;
    tk_mi355x_cortex_stats_t stats{};
    std::string last_response, last_prompt;
    mutable std::mutex stat_mu;
    /* response thread */
    std::thread llm_thread;
    std::mutex llm_mu;
    std::condition_variable llm_cv;
    struct Prompt { bool from_speech; };
    std::deque<Prompt> prompts;
    bool llm_stop = false;
};

static void set_state(tk_cortex_s* c, tk_system_state_e s) {
    if (c->state.exchange(s) != s && c->cb.on_state_change) c->cb.on_state_change(s, c->config.user_data);
}

static tk_error_code_t mkpath(const char* s, const char* dflt, tk_path_t** out) { return tk_path_create_from_string(out, s ? s : dflt); }

/* one LLM turn as cortex_run_llm_inference does it (tk_cortex_main.c:1323-1379): the prompt is the reasoner's context string under a
 * 2048-token budget; the response text is parsed into actions (tk_decision_engine.c:1632) — executing them (TTS, navigation) is outside
 * the hot path, they are counted — and remembered as the system's conversation turn */
static void run_llm(tk_cortex_s* c) {
    set_state(c, TK_STATE_PROCESSING);
    char* ctx = nullptr;
    if (tk_contextual_reasoner_generate_context_string(c->reasoner, &ctx, 2048) != TK_SUCCESS || !ctx) { set_state(c, TK_STATE_IDLE); return; }
    const std::string prompt = ctx;
    (void)tk_contextual_reasoner_free_context_string(ctx);
    static const bool debug = [] { const char* e = getenv("TK_MI355X_DEBUG"); return e && e[0] == '1'; }(); /* failures of the response thread on stderr */
    if (tk_llm_runner_prepare_generation(c->runner, prompt.c_str(), false) != TK_SUCCESS) {
        if (debug) fprintf(stderr, "tk_cortex: prepare_generation failed: %s\n", tk_error_get_detail());
        set_state(c, TK_STATE_IDLE);
        return;
    }
    set_state(c, TK_STATE_RESPONDING);
    std::string resp;
    int n = 0;
    for (; n < c->max_tokens && !c->stop.load(); ++n) {
        const char* p = tk_llm_runner_generate_next_token(c->runner);
        if (!p && debug) /* NULL is also the normal end (EOS, context full): the thread's last error detail is printed for what it is worth, it may predate this response */
            fprintf(stderr, "tk_cortex: generate_next_token returned NULL after %d tokens (EOS / end of context, or a failure; last error detail of this thread: %s)\n", n, tk_error_get_detail());
        if (!p || p == TK_TOOL_CALL_TOKEN) break;
        resp += p;
    }
    tk_llm_response_t* parsed = nullptr;
    const bool ok = tk_decision_engine_parse_llm_response_text(resp.c_str(), &parsed) == TK_SUCCESS;
    const uint64_t n_actions = ok && parsed ? parsed->action_count : 0;
    tk_decision_engine_free_response(&parsed);
    (void)tk_contextual_reasoner_add_conversation_turn(c->reasoner, false, resp.c_str(), 1.0f);
    {
        std::lock_guard<std::mutex> lk(c->stat_mu);
        c->stats.llm_responses++;
        c->stats.llm_tokens += (uint64_t)n;
        c->stats.responses_parsed += ok ? 1 : 0;
        c->stats.actions_parsed += n_actions;
        c->last_response = resp;
        c->last_prompt = prompt;
    }
    set_state(c, TK_STATE_IDLE);
}

/* hand a prompt to the response thread; user speech is never dropped, stale scene descriptions are */
static void post_prompt(tk_cortex_s* c, bool from_speech) {
    std::lock_guard<std::mutex> lk(c->llm_mu);
    if (c->prompts.size() >= TK_CORTEX_PROMPT_QUEUE) {
        for (auto it = c->prompts.begin(); it != c->prompts.end(); ++it)
            if (!it->from_speech) {
                c->prompts.erase(it);
                std::lock_guard<std::mutex> sl(c->stat_mu);
                c->stats.events_dropped++;
                break;
            }
    }
    c->prompts.push_back(tk_cortex_s::Prompt{from_speech});
    c->llm_cv.notify_one();
}

static void llm_worker(tk_cortex_s* c) {
    for (;;) {
        tk_cortex_s::Prompt p;
        {
            std::unique_lock<std::mutex> lk(c->llm_mu);
            c->llm_cv.wait(lk, [&] { return c->llm_stop || !c->prompts.empty(); });
            if (c->llm_stop) return;
            p = c->prompts.front();
            c->prompts.pop_front();
        }
        run_llm(c);
    }
}

extern "C" {

tk_error_code_t tk_cortex_create(tk_cortex_t** out_cortex, const tk_cortex_config_t* config, tk_cortex_callbacks_t callbacks) {
    if (!out_cortex || !config) return TK_ERROR_INVALID_ARGUMENT;
    std::unique_ptr<tk_cortex_s> c(new tk_cortex_s());
    c->config = *config;
    c->cb = callbacks;
    set_state(c.get(), TK_STATE_INITIALIZING);
    const int dev = config->gpu_device_id >= 0 ? config->gpu_device_id : 0;
    tk_error_code_t rc;
    tk_path_t* p = nullptr;
    auto fail = [&](tk_error_code_t e) { tk_cortex_t* raw = c.release(); tk_cortex_destroy(&raw); return e; };
    const tk_model_paths_t& mp = config->model_paths;
    if (!mp.llm_model && !mp.object_detection_model && !mp.depth_estimation_model && !mp.asr_model && !mp.vad_model) {
        /* no model at all (tests/tk_cortex_full_test.c:20-34: "All paths are NULL", device -1): a cortex whose only subsystem is the contextual
         * reasoner.  No GPU is touched; injected frames / audio are refused (TK_ERROR_INVALID_STATE). */
        tk_context_config_t cc{100, 20, 0.3f, 0.95f, 100}; /* tk_cortex_main.c:835-841 */
        if ((rc = tk_contextual_reasoner_create(&c->reasoner, &cc)) != TK_SUCCESS) return fail(rc);
        set_state(c.get(), TK_STATE_IDLE);
        *out_cortex = c.release();
        return TK_SUCCESS;
    }

    if ((rc = mkpath(config->model_paths.object_detection_model, "synthetic://yolov8n?seed=5&cls_bias=-0.45", &p)) != TK_SUCCESS) return fail(rc);
    {
        /* tk_cortex_main.c:774-782: confidence 0.5, at most 20 objects, focal lengths left at zero; the depth model is optional here (the
         * reference's pipeline also only warns when it cannot load it, tk_vision_pipeline.c:395-401) */
        tk_path_t* pd = nullptr;
        if (config->model_paths.depth_estimation_model && config->model_paths.depth_estimation_model[0] &&
            (rc = tk_path_create_from_string(&pd, config->model_paths.depth_estimation_model)) != TK_SUCCESS) { tk_path_destroy(&p); return fail(rc); }
        tk_vision_pipeline_config_t vc{};
        vc.backend = TK_VISION_BACKEND_ROCM; vc.gpu_device_id = dev; vc.object_detection_model_path = p; vc.depth_estimation_model_path = pd;
        vc.object_confidence_threshold = 0.5f; vc.max_detected_objects = 20;
        rc = tk_vision_pipeline_create(&c->vis, &vc);
        tk_path_destroy(&p);
        if (pd) tk_path_destroy(&pd);
        if (rc != TK_SUCCESS) return fail(rc);
    }

    if ((rc = mkpath(config->model_paths.asr_model, "synthetic://whisper-tiny.en?seed=6", &p)) != TK_SUCCESS) return fail(rc);
    tk_asr_whisper_config_t ac{};
    ac.model_path = p; ac.language = config->user_language ? config->user_language : "en"; ac.sample_rate = 16000; ac.n_threads = 4; ac.max_context = 16384;
    rc = tk_asr_whisper_create(&c->asr, &ac);
    tk_path_destroy(&p);
    if (rc != TK_SUCCESS) return fail(rc);

    if ((rc = mkpath(config->model_paths.vad_model, "synthetic://vad?seed=7", &p)) != TK_SUCCESS) return fail(rc);
    tk_vad_silero_config_t vc{};
    vc.model_path = p; vc.sample_rate = 16000; vc.threshold = 0.8f; vc.min_silence_duration_ms = 500.0f; /* tk_cortex_main.c:881-882 */
    rc = tk_vad_silero_create(&c->vad, &vc);
    tk_path_destroy(&p);
    if (rc != TK_SUCCESS) return fail(rc);

    tk_model_loader_config_t lc{2, 1};
    if ((rc = tk_model_loader_create(&c->loader, &lc)) != TK_SUCCESS) return fail(rc);
    if ((rc = mkpath(config->model_paths.llm_model, "synthetic://mistral-7b?seed=4", &p)) != TK_SUCCESS) return fail(rc);
    tk_model_load_params_t lp{};
    lp.model_path = p; lp.model_type = TK_MODEL_FORMAT_GGUF; lp.gpu_layers = 99;
    rc = tk_model_loader_load_model(c->loader, &lp, &c->llm_model);
    tk_path_destroy(&p);
    if (rc != TK_SUCCESS) return fail(rc);
    tk_llm_config_t rcfg{4096, NULL, 0}; /* n_ctx 4096: tk_runner_lifecycle.c:48 */
    if ((rc = tk_llm_runner_create(&c->runner, c->llm_model, &rcfg)) != TK_SUCCESS) return fail(rc);
    tk_context_config_t cc{100, 20, 0.3f, 0.95f, 100}; /* tk_cortex_main.c:835-841 */
    if ((rc = tk_contextual_reasoner_create(&c->reasoner, &cc)) != TK_SUCCESS) return fail(rc);

    c->llm_thread = std::thread(llm_worker, c.get());
    set_state(c.get(), TK_STATE_IDLE);
    *out_cortex = c.release();
    return TK_SUCCESS;
}

void tk_cortex_destroy(tk_cortex_t** cortex) {
    if (!cortex || !*cortex) return;
    tk_cortex_s* c = *cortex;
    c->stop.store(true);
    c->cv.notify_all();
    set_state(c, TK_STATE_SHUTDOWN);
    {
        std::lock_guard<std::mutex> lk(c->llm_mu);
        c->llm_stop = true;
    }
    c->llm_cv.notify_all();
    if (c->llm_thread.joinable()) c->llm_thread.join(); /* run_llm leaves its token loop as soon as `stop` is set */
    if (c->runner) tk_llm_runner_destroy(&c->runner);
    if (c->reasoner) tk_contextual_reasoner_destroy(&c->reasoner);
    if (c->loader) { if (c->llm_model) (void)tk_model_loader_unload_model(c->loader, &c->llm_model); tk_model_loader_destroy(&c->loader); }
    if (c->vis) tk_vision_pipeline_destroy(&c->vis);
    if (c->asr) tk_asr_whisper_destroy(&c->asr);
    if (c->vad) tk_vad_silero_destroy(&c->vad);
    delete c;
    *cortex = nullptr;
}

tk_error_code_t tk_cortex_run(tk_cortex_t* c) {
    if (!c) return TK_ERROR_INVALID_ARGUMENT;
    if (c->state.load() == TK_STATE_UNINITIALIZED || c->state.load() == TK_STATE_FATAL_ERROR) return TK_ERROR_INVALID_STATE;
    while (!c->stop.load()) {
        CortexEvent ev;
        {
            std::unique_lock<std::mutex> lk(c->mu);
            c->cv.wait(lk, [&] { return c->stop.load() || !c->queue.empty(); });
            if (c->stop.load()) break;
            ev = c->queue.front();
            c->queue.pop_front();
        }
        if (ev.kind == CortexEvent::VIDEO) {
            std::vector<uint8_t> data;
            tk_video_frame_t f;
            {
                std::lock_guard<std::mutex> lk(c->mu); /* copy under the lock, like tk_cortex_main.c:1162-1167 */
                f = c->ring[ev.slot].f;
                data = c->ring[ev.slot].data;
            }
            f.data = data.data();
            /* cortex_process_vision_input (tk_cortex_main.c:1186-1222): the ENVIRONMENT_AWARENESS preset through the vision pipeline, the
             * result (objects with attributes and, when a depth model is configured, fused distances) into the reasoner's snapshot */
            tk_vision_result_t* vr = nullptr;
            const tk_vision_analysis_flags_t flags = TK_VISION_ANALYZE_OBJECT_DETECTION | TK_VISION_ANALYZE_DEPTH_ESTIMATION |
                                                     TK_VISION_ANALYZE_FUSION_DISTANCE | TK_VISION_ANALYZE_NAVIGATION_CUES;
            if (tk_vision_pipeline_process_frame(c->vis, &f, flags, nullptr, 0, &vr) != TK_SUCCESS || !vr) {
                std::lock_guard<std::mutex> lk(c->stat_mu); /* the frame is lost: counted, like an event the queue refused */
                c->stats.events_dropped++;
                continue;
            }
            const size_t n = vr->object_count;
            (void)tk_contextual_reasoner_update_vision_context(c->reasoner, vr);
            tk_vision_result_destroy(&vr);
            {
                std::lock_guard<std::mutex> lk(c->stat_mu);
                c->stats.frames_processed++;
                if (n > 0) c->stats.frames_with_objects++;
            }
            if (n > 0) post_prompt(c, false); /* SIGNIFICANT_VISION_CHANGE: every frame with >= 1 detection (tk_cortex_main.c:1224-1237) */
        } else {
            (void)tk_contextual_reasoner_add_conversation_turn(c->reasoner, true, ev.text.c_str(), 0.9f); /* final transcription (:1081) */
            post_prompt(c, true);
        }
    }
    return TK_SUCCESS;
}

tk_error_code_t tk_cortex_stop(tk_cortex_t* c) {
    if (!c) return TK_ERROR_INVALID_ARGUMENT;
    c->stop.store(true);
    c->cv.notify_all();
    return TK_SUCCESS;
}

static bool enqueue(tk_cortex_s* c, CortexEvent&& ev) {
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->queue.size() >= TK_CORTEX_QUEUE_CAP) {
        std::lock_guard<std::mutex> sl(c->stat_mu);
        c->stats.events_dropped++;
        return false;
    }
    c->queue.push_back(std::move(ev));
    c->cv.notify_one();
    return true;
}

tk_contextual_reasoner_t* tk_cortex_get_contextual_reasoner(tk_cortex_t* cortex) { return cortex ? cortex->reasoner : NULL; } /* tk_cortex_main.c:743-748 */

tk_error_code_t tk_cortex_inject_video_frame(tk_cortex_t* c, const tk_video_frame_t* frame) {
    if (!c || !frame || !frame->data || frame->width == 0 || frame->height == 0) return TK_ERROR_INVALID_ARGUMENT;
    if (!c->vis) return TK_ERROR_INVALID_STATE; /* a cortex created without models */
    const uint32_t bpp = frame->format == TK_PIXEL_FORMAT_RGBA8 ? 4 : 3;
    const uint32_t stride = frame->stride ? frame->stride : frame->width * bpp;
    int slot;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        slot = c->ring_next;
        c->ring_next = (c->ring_next + 1) % TK_CORTEX_VIDEO_RING;
        c->ring[slot].f = *frame;
        c->ring[slot].f.stride = stride;
        c->ring[slot].data.assign(frame->data, frame->data + (size_t)stride * frame->height);
    }
    CortexEvent ev;
    ev.kind = CortexEvent::VIDEO;
    ev.slot = slot;
    return enqueue(c, std::move(ev)) ? TK_SUCCESS : TK_ERROR_TASK_QUEUE_FULL;
}

struct VadCtx { tk_cortex_s* c; bool started, ended; };
static void on_vad(tk_vad_silero_event_e e, void* u) {
    VadCtx* v = (VadCtx*)u;
    if (e == TK_VAD_EVENT_SPEECH_STARTED) v->started = true;
    else v->ended = true;
}

tk_error_code_t tk_cortex_inject_audio_frame(tk_cortex_t* c, const int16_t* audio_data, size_t frame_count) {
    if (!c || !audio_data) return TK_ERROR_INVALID_ARGUMENT;
    if (!c->vad || !c->asr) return TK_ERROR_INVALID_STATE; /* a cortex created without models */
    std::lock_guard<std::mutex> lk(c->audio_mu);
    VadCtx v{c, false, false};
    tk_error_code_t rc = tk_vad_silero_process_audio_with_events(c->vad, audio_data, frame_count, on_vad, &v);
    if (rc != TK_SUCCESS) return rc;
    if (v.started) { c->in_speech = true; set_state(c, TK_STATE_LISTENING); }
    if (c->in_speech) c->speech.insert(c->speech.end(), audio_data, audio_data + frame_count);
    if (v.ended && c->in_speech) {
        c->in_speech = false;
        tk_asr_whisper_result_t* r = nullptr;
        const size_t n = c->speech.size() < 480000 ? c->speech.size() : 480000;
        rc = tk_asr_whisper_process_audio(c->asr, c->speech.data(), n, true, &r);
        c->speech.clear();
        if (rc != TK_SUCCESS) return rc;
        CortexEvent ev;
        ev.kind = CortexEvent::SPEECH;
        ev.text = r && r->text ? r->text : "";
        tk_asr_whisper_free_result(&r);
        {
            std::lock_guard<std::mutex> sl(c->stat_mu);
            c->stats.speech_segments++;
        }
        if (!enqueue(c, std::move(ev))) return TK_ERROR_TASK_QUEUE_FULL;
    }
    return TK_SUCCESS;
}

tk_error_code_t tk_cortex_inject_sensor_event(tk_cortex_t* c, const tk_sensor_event_t* event) {
    if (!c || !event) return TK_ERROR_INVALID_ARGUMENT;
    return TK_SUCCESS; /* sensor fusion is outside the hot path */
}

tk_error_code_t tk_cortex_get_state(const tk_cortex_t* c, tk_system_state_e* out_state) {
    if (!c || !out_state) return TK_ERROR_INVALID_ARGUMENT;
    *out_state = (tk_system_state_e)c->state.load();
    return TK_SUCCESS;
}

void tk_mi355x_cortex_get_stats(const tk_cortex_t* c, tk_mi355x_cortex_stats_t* out) {
    if (!c || !out) return;
    std::lock_guard<std::mutex> lk(c->stat_mu);
    *out = c->stats;
}

size_t tk_mi355x_cortex_last_response(const tk_cortex_t* c, char* buf, size_t cap) {
    if (!c) return 0;
    std::lock_guard<std::mutex> lk(c->stat_mu);
    if (buf && cap) {
        size_t n = c->last_response.size() < cap - 1 ? c->last_response.size() : cap - 1;
        memcpy(buf, c->last_response.data(), n);
        buf[n] = 0;
    }
    return c->last_response.size();
}

size_t tk_mi355x_cortex_last_prompt(const tk_cortex_t* c, char* buf, size_t cap) {
    if (!c) return 0;
    std::lock_guard<std::mutex> lk(c->stat_mu);
    if (buf && cap) {
        size_t n = c->last_prompt.size() < cap - 1 ? c->last_prompt.size() : cap - 1;
        memcpy(buf, c->last_prompt.data(), n);
        buf[n] = 0;
    }
    return c->last_prompt.size();
}

void tk_mi355x_cortex_set_max_response_tokens(tk_cortex_t* c, int n) {
    if (c && n > 0) c->max_tokens = n;
}

} /* extern "C" */
