/*
 * tk_abi_vision.cpp — tk_object_detector_* / tk_preprocessor_* on the HIP vision engine.
 * Call sequence mirrored from src/vision/tk_object_detector.c:182-219 (detect = preprocess ->
 * inference -> postprocess -> caller-freed malloc'd array) and :83-178 (create/destroy).
 */
#include <stdlib.h>
#include <string.h>

#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../vision/tk_onnx_weights.h"
#include "../vision/tk_vision_engine.h"
#include "tk/tk_mi355x_ext.h"
#include "tk/tk_vision.h"
#include "tk/tk_depth.h"
#include "../vision/tk_fusion.h"

#include <string>
#include <sys/stat.h>
/* a registry key names a FILE, not a path: a path rewritten while an older handle is alive (size, modification time or inode differ) is another model */
static std::string file_identity(const std::string& path) {
    struct stat st;
    if (path.compare(0, 12, "synthetic://") == 0 || stat(path.c_str(), &st) != 0) return path;
    return path + "|" + std::to_string((long long)st.st_size) + "|" + std::to_string((long long)st.st_mtim.tv_sec) + "." + std::to_string((long long)st.st_mtim.tv_nsec) + "|" +
           std::to_string((unsigned long long)st.st_ino);
}

/* ---- per-model-file registry: every tk_object_detector_t opened on the same file / device / geometry shares ONE set of weights
 * and ONE batched engine.  The reference's API is one frame per call and per handle (src/vision/tk_object_detector.c:182-219); K cortex
 * handles that each own a detector at batch 1 ran K small networks side by side.  Here a call enqueues its frame, a scheduler thread coalesces
 * the frames that are waiting (same geometry and thresholds) into one batched detect() — the way csrc/llm/tk_llm_batcher does for runners — and
 * every caller gets exactly what it would get alone: the engine's arithmetic is per frame (tests/test_vision_gpu.py: batch == singles).
 * The per-box attributes of tk_vision_pipeline_process_frame run inside the same job, while the batch's frames are still on the device. */
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <thread>

#define TK_DET_SHARED_MAX_BATCH 32

struct DetReq {
    const uint8_t* frame = nullptr;
    uint32_t w = 0, h = 0, stride = 0, bpp = 3;
    float conf = 0.5f, iou = 0.5f;
    bool fast = false;           /* tk_mi355x_detector_set_fast_contraction: requests of one kind share a job */
    bool want_attr = false;      /* classify the first `max_objects` boxes (0 = all) while the frame is resident */
    uint32_t max_objects = 0;
    std::vector<TkDetection> dets;
    std::vector<int32_t> color, door;
    bool have_attr = false, ok = false, done = false;
    std::string err;
};

/* the reference's box conversion (detector space -> frame pixels, truncated): one definition for results and attribute rectangles */
static inline void box_to_rect(const TkDetection& v, float sx, float sy, int32_t* r) {
    r[0] = (int)(v.x1 * sx); r[1] = (int)(v.y1 * sy); r[2] = (int)((v.x2 - v.x1) * sx); r[3] = (int)((v.y2 - v.y1) * sy);
}

struct SharedDetector {
    std::string key;
    TkYoloModel model;
    int in_w = 640, in_h = 640;
    std::unique_ptr<TkDetector> eng; /* grows with the number of handles (1, 2, 4 ... TK_DET_SHARED_MAX_BATCH frames per call) */
    std::mutex mu;
    std::condition_variable cv_req, cv_done;
    std::deque<DetReq*> q;
    std::thread th;
    bool stop = false;
    int handles = 0;
    uint64_t n_batches = 0, n_frames = 0, widest = 0;

    ~SharedDetector() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_req.notify_all();
        if (th.joinable()) th.join();
    }
    std::mutex eng_mu; /* held while a job runs on `eng` and while `eng` is replaced */
    /* called when a handle joins (under the registry lock): the engine is sized for the handles there are — 1, 2, 4 ... MAX frames / utterances per
     * job — at CREATE time, so that nothing allocates or frees device memory while jobs and other streams' graph captures run */
    bool grow_for_handles(std::string* err) {
        int want = 1;
        while (want < handles && want < TK_DET_SHARED_MAX_BATCH) want *= 2;
        std::lock_guard<std::mutex> lk(eng_mu);
        return ensure_engine(want, err);
    }
    bool ensure_engine(int want, std::string* err) {
        if (eng && eng->max_batch >= want) return true;
        std::unique_ptr<TkDetector> nd(new TkDetector());
        if (!nd->init(&model, in_w, in_h, want)) { *err = nd->error; return false; }
        eng.swap(nd);
        return true;
    }
    void run() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_req.wait(lk, [&] { return stop || !q.empty(); });
            if (stop) break;
            /* give the other handles' callers a moment to arrive: a few hundred microseconds against a network of ~1.5 ms per frame */
            if (handles > 1) {
                const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(400);
                while ((int)q.size() < (handles < TK_DET_SHARED_MAX_BATCH ? handles : TK_DET_SHARED_MAX_BATCH) && !stop)
                    if (cv_req.wait_until(lk, until) == std::cv_status::timeout) break;
                if (stop) break;
            }
            std::vector<DetReq*> job;
            DetReq* first = q.front();
            for (auto it = q.begin(); it != q.end() && (int)job.size() < TK_DET_SHARED_MAX_BATCH;) {
                DetReq* r = *it;
                if (r->w == first->w && r->h == first->h && r->stride == first->stride && r->bpp == first->bpp && r->conf == first->conf && r->iou == first->iou && r->fast == first->fast) {
                    job.push_back(r);
                    it = q.erase(it);
                } else ++it;
            }
            int cap = 1;
            while (cap < (int)job.size() || (cap < handles && cap < TK_DET_SHARED_MAX_BATCH)) cap *= 2;
            lk.unlock();
            std::string err;
            std::unique_lock<std::mutex> el(eng_mu);
            bool ok = ensure_engine(cap, &err); /* a no-op after grow_for_handles(): the engine already takes this many */
            std::vector<std::vector<TkDetection>> out;
            if (ok) {
                std::vector<const uint8_t*> ptrs(job.size());
                for (size_t i = 0; i < job.size(); ++i) ptrs[i] = job[i]->frame;
                eng->conf = first->conf;
                eng->iou = first->iou;
                eng->fast = first->fast;
                ok = eng->detect((int)job.size(), ptrs.data(), first->w, first->h, first->stride, first->bpp, &out);
                if (!ok) err = eng->error;
            }
            for (size_t i = 0; i < job.size(); ++i) {
                DetReq* r = job[i];
                r->ok = ok;
                if (!ok) { r->err = err; continue; }
                r->dets.swap(out[i]);
                size_t n = r->dets.size();
                if (r->max_objects && n > r->max_objects) n = r->max_objects;
                if (r->want_attr && n > 0) {
                    std::vector<int32_t> rects(4 * n);
                    const float sx = (float)r->w / (float)in_w, sy = (float)r->h / (float)in_h;
                    for (size_t k = 0; k < n; ++k) box_to_rect(r->dets[k], sx, sy, &rects[4 * k]);
                    r->color.assign(n, -1);
                    r->door.assign(n, 0);
                    r->have_attr = eng->classify_boxes((int)i, (int)n, rects.data(), r->color.data(), r->door.data());
                }
            }
            el.unlock();
            lk.lock();
            n_batches++;
            n_frames += job.size();
            if (job.size() > widest) widest = job.size();
            for (DetReq* r : job) r->done = true;
            cv_done.notify_all();
        }
        /* whoever is still waiting is told so */
        for (DetReq* r : q) { r->ok = false; r->err = "the detector was destroyed"; r->done = true; }
        q.clear();
        cv_done.notify_all();
    }
    void submit(DetReq* r) {
        std::unique_lock<std::mutex> lk(mu);
        q.push_back(r);
        cv_req.notify_all();
        cv_done.wait(lk, [&] { return r->done; });
    }
};

static std::mutex g_det_mu;
static std::map<std::string, std::weak_ptr<SharedDetector>> g_det_registry;

struct tk_object_detector_s {
    std::shared_ptr<SharedDetector> sh; /* weights + the coalescing engine, shared by every handle of the same file */
    std::unique_ptr<TkDetector> own;    /* a private engine on the shared weights: only for the batch / raw-tensor entry points (tk_mi355x_detector_*) */
    std::vector<const char*> labels;
    size_t class_count = 0;
    int in_w = 640, in_h = 640;
    float conf = 0.5f, iou = 0.5f;
    bool fast = false; /* tk_mi355x_detector_set_fast_contraction */
    std::vector<std::vector<TkDetection>> last;
    ~tk_object_detector_s() {
        own.reset(); /* before the weights it reads */
        if (sh) { std::lock_guard<std::mutex> lk(sh->mu); sh->handles--; }
    }
    TkDetector* priv(int max_batch, std::string* err) {
        if (own && own->max_batch >= max_batch) return own.get();
        std::unique_ptr<TkDetector> nd(new TkDetector());
        if (!nd->init(&sh->model, in_w, in_h, max_batch)) { *err = nd->error; return nullptr; }
        own.swap(nd);
        return own.get();
    }
};

static tk_error_code_t vfail(tk_error_code_t code, const std::string& why) {
    tk_error_set_detail("%s", why.c_str());
    return code;
}

static double query_param(const std::string& p, const char* key, double dflt) {
    size_t k = p.find(std::string(key) + "=");
    if (k == std::string::npos) return dflt;
    return strtod(p.c_str() + k + strlen(key) + 1, nullptr);
}

extern "C" {

tk_error_code_t tk_object_detector_create(tk_object_detector_t** out_detector, const tk_object_detector_config_t* config) {
    if (!out_detector || !config || !config->model_path || !config->model_path->path_str) return TK_ERROR_INVALID_ARGUMENT;
    if (config->class_count == 0 || !config->class_labels) return vfail(TK_ERROR_INVALID_ARGUMENT, "class_labels / class_count missing");
    if (config->backend == TK_VISION_BACKEND_CPU) return vfail(TK_ERROR_BACKEND_NOT_SUPPORTED, "this library is the MI355X path: there is no CPU backend");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return vfail(TK_ERROR_GPU_DEVICE_NOT_FOUND, "no HIP device visible");
    const int dev = config->gpu_device_id >= 0 ? config->gpu_device_id : 0;
    if (dev >= ndev) return vfail(TK_ERROR_INVALID_ARGUMENT, "gpu_device_id out of range");
    std::unique_ptr<tk_object_detector_s> d(new tk_object_detector_s());
    d->class_count = config->class_count;
    d->labels.assign(config->class_labels, config->class_labels + config->class_count);
    d->in_w = config->input_width ? (int)config->input_width : 640;
    d->in_h = config->input_height ? (int)config->input_height : 640;
    d->conf = config->confidence_threshold;
    d->iou = config->iou_threshold;
    const std::string path = config->model_path->path_str;
    const std::string key = file_identity(path) + "|dev" + std::to_string(dev) + "|nc" + std::to_string(config->class_count) + "|" + std::to_string(d->in_w) + "x" + std::to_string(d->in_h);
    {
        /* find-or-load under the registry lock: K handles created at once load the file once (the reference's model loader does the same for
         * LLM files, src/ai_models/tk_model_loader.c:199-294) */
        std::lock_guard<std::mutex> lk(g_det_mu);
        std::shared_ptr<SharedDetector> sh = g_det_registry[key].lock();
        if (!sh) {
            sh.reset(new SharedDetector());
            sh->key = key;
            sh->in_w = d->in_w;
            sh->in_h = d->in_h;
            if (!sh->model.init(dev, (int)config->class_count)) return vfail(TK_ERROR_MODEL_LOAD_FAILED, sh->model.error);
            if (path.compare(0, 12, "synthetic://") == 0) {
                if (!sh->model.fill_synthetic((uint64_t)query_param(path, "seed", 5), (float)query_param(path, "cls_bias", -4.0)))
                    return vfail(TK_ERROR_MODEL_LOAD_FAILED, sh->model.error);
            } else if (path.size() > 5 && path.compare(path.size() - 5, 5, ".onnx") == 0) {
                /* the reference's detector file: its Conv initialisers feed this path's own YOLOv8n graph (no ONNX Runtime) */
                if (!sh->model.load_onnx(path.c_str())) return vfail(TK_ERROR_MODEL_LOAD_FAILED, sh->model.error);
            } else if (!sh->model.load_file(path.c_str())) {
                return vfail(TK_ERROR_MODEL_LOAD_FAILED, sh->model.error);
            }
            std::string err;
            if (!sh->ensure_engine(1, &err)) return vfail(TK_ERROR_GPU_MEMORY, err);
            SharedDetector* raw = sh.get();
            sh->th = std::thread([raw] { raw->run(); });
            g_det_registry[key] = sh;
        }
        { std::lock_guard<std::mutex> hl(sh->mu); sh->handles++; }
        d->sh = sh;
        std::string gerr;
        if (!sh->grow_for_handles(&gerr)) return vfail(TK_ERROR_GPU_MEMORY, gerr); /* (d's destructor takes the handle back) */
    }
    *out_detector = d.release();
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_onnx_probe(const char* path, int32_t* n_convs, int64_t* n_params) {
    if (!path) return TK_ERROR_INVALID_ARGUMENT;
    TkOnnxWeights ox;
    if (!ox.load(path)) return vfail(TK_ERROR_MODEL_LOAD_FAILED, ox.error);
    int64_t p = 0;
    for (const auto& c : ox.convs) p += (int64_t)c.w.size() + (int64_t)c.b.size();
    if (n_convs) *n_convs = (int32_t)ox.convs.size();
    if (n_params) *n_params = p;
    /* YOLOv8n check: the 63 convolutions of the graph, in order, plus optionally the DFL projection */
    const std::vector<TkConvSpec> specs = TkYoloV8n<int>::specs(ox.convs.size() >= 63 ? ox.convs[62].cout : TK_YOLO_NC);
    size_t n = ox.convs.size();
    if (n == specs.size() + 1 && ox.convs.back().cout == 1 && ox.convs.back().cin == TK_YOLO_REG_MAX) --n;
    if (n != specs.size()) return vfail(TK_ERROR_MODEL_VERIFICATION_FAILED, "not a YOLOv8n graph: " + std::to_string(ox.convs.size()) + " Conv nodes");
    for (size_t i = 0; i < n; ++i)
        if (ox.convs[i].cout != specs[i].cout || ox.convs[i].cin != specs[i].cin || ox.convs[i].kh != specs[i].k || ox.convs[i].kw != specs[i].k)
            return vfail(TK_ERROR_MODEL_VERIFICATION_FAILED, "Conv " + std::to_string(i) + " does not match the YOLOv8n graph");
    return TK_SUCCESS;
}

void tk_object_detector_destroy(tk_object_detector_t** detector) {
    if (!detector || !*detector) return;
    delete *detector;
    *detector = nullptr;
}

tk_error_code_t tk_mi355x_detector_set_max_batch(tk_object_detector_t* d, int max_batch) {
    if (!d || max_batch < 1) return TK_ERROR_INVALID_ARGUMENT;
    std::string err;
    d->own.reset();
    if (!d->priv(max_batch, &err)) return vfail(TK_ERROR_GPU_MEMORY, err);
    return TK_SUCCESS;
}

void tk_mi355x_detector_share_stats(const tk_object_detector_t* d, uint64_t* handles, uint64_t* batches, uint64_t* frames, uint64_t* widest) {
    if (!d || !d->sh) return;
    std::lock_guard<std::mutex> lk(d->sh->mu);
    if (handles) *handles = (uint64_t)d->sh->handles;
    if (batches) *batches = d->sh->n_batches;
    if (frames) *frames = d->sh->n_frames;
    if (widest) *widest = d->sh->widest;
}

static tk_error_code_t to_results(tk_object_detector_s* d, const std::vector<TkDetection>& v, const tk_video_frame_t* f,
                                  tk_detection_result_t** out, size_t* count) {
    *count = v.size();
    *out = (tk_detection_result_t*)malloc((v.size() ? v.size() : 1) * sizeof(tk_detection_result_t));
    if (!*out) return TK_ERROR_OUT_OF_MEMORY;
    const float sx = (float)f->width / (float)d->in_w, sy = (float)f->height / (float)d->in_h;
    for (size_t i = 0; i < v.size(); ++i) {
        tk_detection_result_t& r = (*out)[i];
        r.class_id = (uint32_t)v[i].cls;
        r.label = v[i].cls >= 0 && (size_t)v[i].cls < d->class_count ? d->labels[v[i].cls] : NULL;
        r.confidence = v[i].score;
        int32_t rc4[4];
        box_to_rect(v[i], sx, sy, rc4);
        r.bbox.x = rc4[0]; r.bbox.y = rc4[1]; r.bbox.w = rc4[2]; r.bbox.h = rc4[3];
    }
    return TK_SUCCESS;
}

static int bytes_per_pixel(const tk_video_frame_t* f) { return f->format == TK_PIXEL_FORMAT_RGBA8 ? 4 : 3; }

tk_error_code_t tk_mi355x_detector_detect_batch(tk_object_detector_t* d, int n, const tk_video_frame_t* frames, tk_detection_result_t** out_results,
                                                size_t* out_counts) {
    if (!d || !frames || !out_results || !out_counts || n <= 0) return TK_ERROR_INVALID_ARGUMENT;
    std::vector<const uint8_t*> ptrs(n);
    /* stride 0 = tightly packed rows, as in the other entry points (the reference detector ignores `stride` and assumes width * 3,
     * tk_object_detector.c:235, so callers that leave it unset must keep working) */
    const uint32_t bpp = (uint32_t)bytes_per_pixel(&frames[0]);
    auto pitch = [bpp](const tk_video_frame_t& f) { return f.stride ? f.stride : f.width * bpp; };
    for (int i = 0; i < n; ++i) {
        if (!frames[i].data) return TK_ERROR_INVALID_ARGUMENT;
        if (frames[i].width != frames[0].width || frames[i].height != frames[0].height || frames[i].format != frames[0].format ||
            pitch(frames[i]) != pitch(frames[0]))
            return vfail(TK_ERROR_INVALID_ARGUMENT, "frames of one batch must share geometry");
        ptrs[i] = frames[i].data;
    }
    if (n == 1 && !d->own) {
        /* the reference's one-frame call: through the shared engine, coalesced with the other handles' frames */
        DetReq r;
        r.frame = frames[0].data; r.w = frames[0].width; r.h = frames[0].height; r.stride = pitch(frames[0]); r.bpp = bpp; r.conf = d->conf; r.iou = d->iou; r.fast = d->fast;
        if (r.w < 2 || r.h < 2 || r.stride < r.w * bpp) return vfail(TK_ERROR_INFERENCE_FAILED, "frame geometry invalid (need w,h >= 2 and stride >= w*bpp)");
        d->sh->submit(&r);
        if (!r.ok) return vfail(TK_ERROR_INFERENCE_FAILED, r.err);
        d->last.assign(1, std::vector<TkDetection>());
        d->last[0].swap(r.dets);
    } else {
        std::string err;
        TkDetector* e = d->priv(n, &err); /* the handle's private engine, grown when this batch is wider than it */
        if (!e) return vfail(TK_ERROR_GPU_MEMORY, err);
        e->conf = d->conf;
        e->iou = d->iou;
        e->fast = d->fast;
        if (!e->detect(n, ptrs.data(), frames[0].width, frames[0].height, pitch(frames[0]), bpp, &d->last)) return vfail(TK_ERROR_INFERENCE_FAILED, e->error);
    }
    for (int i = 0; i < n; ++i) {
        tk_error_code_t rc = to_results(d, d->last[i], &frames[i], &out_results[i], &out_counts[i]);
        if (rc != TK_SUCCESS) return rc;
    }
    return TK_SUCCESS;
}

tk_error_code_t tk_object_detector_detect(tk_object_detector_t* detector, const tk_video_frame_t* video_frame, tk_detection_result_t** out_results,
                                          size_t* out_result_count) {
    if (!detector || !video_frame || !video_frame->data || !out_results || !out_result_count) return TK_ERROR_INVALID_ARGUMENT;
    return tk_mi355x_detector_detect_batch(detector, 1, video_frame, out_results, out_result_count);
}

void tk_object_detector_free_results(tk_detection_result_t** results) {
    if (!results || !*results) return;
    free(*results);
    *results = NULL;
}

void tk_object_detector_update_thresholds(tk_object_detector_t* detector, float confidence_threshold, float iou_threshold) {
    if (!detector) return;
    detector->conf = confidence_threshold;
    detector->iou = iou_threshold;
}

tk_error_code_t tk_mi355x_detector_set_fast_contraction(tk_object_detector_t* detector, int on) {
    if (!detector) return TK_ERROR_INVALID_ARGUMENT;
    detector->fast = on != 0;
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_detector_forward_raw(tk_object_detector_t* d, int batch, const float* nhwc, float* raw_out, size_t raw_floats) {
    if (!d || !nhwc || !raw_out) return TK_ERROR_INVALID_ARGUMENT;
    std::vector<float> raw;
    std::string err;
    TkDetector* e = d->priv(batch > 1 ? batch : 1, &err);
    if (!e) return vfail(TK_ERROR_GPU_MEMORY, err);
    e->conf = d->conf;
    e->iou = d->iou;
    e->fast = d->fast;
    if (!e->forward_tensor(batch, nhwc, &raw)) return vfail(TK_ERROR_INFERENCE_FAILED, e->error);
    if (raw.size() > raw_floats) return vfail(TK_ERROR_BUFFER_TOO_SMALL, "raw_out too small");
    memcpy(raw_out, raw.data(), raw.size() * 4);
    if (!e->fetch(batch, &d->last)) return vfail(TK_ERROR_INFERENCE_FAILED, e->error); /* for last_boxes */
    return TK_SUCCESS;
}

tk_error_code_t tk_mi355x_detector_forward_graph(tk_object_detector_t* d, int batch, const float* nchw, float* out, size_t out_floats) {
    if (!d || !nchw || !out) return TK_ERROR_INVALID_ARGUMENT;
    std::vector<float> o;
    std::string err;
    TkDetector* e = d->priv(batch > 1 ? batch : 1, &err);
    if (!e) return vfail(TK_ERROR_GPU_MEMORY, err);
    e->conf = d->conf;
    e->iou = d->iou;
    e->fast = d->fast;
    if (!e->forward_graph(batch, nchw, &o)) return vfail(TK_ERROR_INFERENCE_FAILED, e->error);
    if (o.size() > out_floats) return vfail(TK_ERROR_BUFFER_TOO_SMALL, "out too small");
    memcpy(out, o.data(), o.size() * 4);
    if (!e->fetch(batch, &d->last)) return vfail(TK_ERROR_INFERENCE_FAILED, e->error); /* for last_boxes */
    return TK_SUCCESS;
}

int tk_mi355x_detector_is_graph(const tk_object_detector_t* d) { return d && d->sh && d->sh->model.generic ? 1 : 0; }

int tk_mi355x_detector_anchor_count(const tk_object_detector_t* d) {
    if (!d) return 0;
    if (d->own) return d->own->n_anchors;
    std::lock_guard<std::mutex> lk(d->sh->eng_mu);
    return d->sh->eng ? d->sh->eng->n_anchors : 0;
}

tk_error_code_t tk_mi355x_detector_last_boxes(tk_object_detector_t* d, int frame, float* boxes5, int32_t* cls, int32_t* anchors, int cap, int* count) {
    if (!d || !count || frame < 0 || frame >= (int)d->last.size()) return TK_ERROR_INVALID_ARGUMENT;
    const auto& v = d->last[frame];
    *count = (int)v.size();
    for (int i = 0; i < (int)v.size() && i < cap; ++i) {
        if (boxes5) { boxes5[5 * i] = v[i].x1; boxes5[5 * i + 1] = v[i].y1; boxes5[5 * i + 2] = v[i].x2; boxes5[5 * i + 3] = v[i].y2; boxes5[5 * i + 4] = v[i].score; }
        if (cls) cls[i] = v[i].cls;
        if (anchors) anchors[i] = v[i].anchor;
    }
    return TK_SUCCESS;
}

/* ---- stand-alone pre-processor entry (reference: tk_image_preprocessor.c:21) ---- */
tk_error_code_t tk_preprocessor_resize_and_normalize_to_chw(const tk_video_frame_t* frame, float* out_tensor, uint32_t target_width,
                                                            uint32_t target_height, const float mean[3], const float std_dev[3]) {
    if (!frame || !frame->data || !out_tensor || target_width == 0 || target_height == 0 || !mean || !std_dev) return TK_ERROR_INVALID_ARGUMENT;
    if (frame->width < 2 || frame->height < 2) return vfail(TK_ERROR_INVALID_ARGUMENT, "frame must be at least 2x2");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return vfail(TK_ERROR_GPU_DEVICE_NOT_FOUND, "no HIP device visible");
    if (hipSetDevice(tk_mi355x_get_default_device()) != hipSuccess) return vfail(TK_ERROR_GPU_ROCM_ERROR, "hipSetDevice failed");
    const uint32_t bpp = (uint32_t)bytes_per_pixel(frame);
    const uint32_t stride = frame->stride ? frame->stride : frame->width * bpp;
    if (stride < frame->width * bpp) return vfail(TK_ERROR_INVALID_ARGUMENT, "stride smaller than a row");
    const size_t fb = (size_t)stride * frame->height, ob = (size_t)target_width * target_height * 3 * sizeof(float);
    uint8_t* dsrc = nullptr;
    float* ddst = nullptr;
    if (hipMalloc((void**)&dsrc, fb) != hipSuccess || hipMalloc((void**)&ddst, ob) != hipSuccess) {
        if (dsrc) (void)hipFree(dsrc);
        return vfail(TK_ERROR_GPU_MEMORY, "device allocation failed");
    }
    tk_error_code_t rc = TK_SUCCESS;
    TkPreprocessArgs a{};
    a.src = dsrc; a.in_w = frame->width; a.in_h = frame->height; a.in_stride = stride; a.bpp = bpp;
    a.dst = ddst; a.out_w = target_width; a.out_h = target_height; a.nhwc = 0;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean[c]; a.std_dev[c] = std_dev[c]; }
    if (hipMemcpy(dsrc, frame->data, fb, hipMemcpyHostToDevice) != hipSuccess) rc = TK_ERROR_GPU_MEMORY;
    if (rc == TK_SUCCESS) {
        tk_launch_preprocess(a, nullptr);
        if (hipGetLastError() != hipSuccess) rc = TK_ERROR_GPU_KERNEL_LAUNCH;
        else if (hipMemcpy(out_tensor, ddst, ob, hipMemcpyDeviceToHost) != hipSuccess) rc = TK_ERROR_GPU_MEMORY;
    }
    (void)hipFree(dsrc);
    (void)hipFree(ddst);
    return rc;
}

/* ------------------------------------------------------------------ per-box attributes ------ */

static const char* const TK_COLOR_NAMES[9] = {"red", "yellow", "green", "cyan", "blue", "magenta", "black", "white", "gray"};

static tk_error_code_t classify_one(const tk_video_frame_t* frame, const tk_rect_t* bbox, int32_t* color, int32_t* door) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return vfail(TK_ERROR_GPU_DEVICE_NOT_FOUND, "no HIP device visible");
    const int32_t r[4] = {bbox->x, bbox->y, bbox->w, bbox->h};
    std::string err;
    if (!tk_classify_boxes_host(tk_mi355x_get_default_device(), frame->data, frame->width, frame->height, 1, r, color, door, &err))
        return vfail(TK_ERROR_GPU_ROCM_ERROR, err);
    return TK_SUCCESS;
}

tk_error_code_t tk_classify_dominant_color(const tk_video_frame_t* frame, const tk_rect_t* bbox, char** out_color_name) {
    if (!frame || !frame->data || !bbox || !out_color_name) return TK_ERROR_INVALID_ARGUMENT;
    int32_t c = 0;
    tk_error_code_t rc = classify_one(frame, bbox, &c, nullptr);
    if (rc != TK_SUCCESS) return rc;
    *out_color_name = strdup(TK_COLOR_NAMES[c]);
    return *out_color_name ? TK_SUCCESS : TK_ERROR_OUT_OF_MEMORY;
}

tk_error_code_t tk_classify_door_state(const tk_video_frame_t* frame, const tk_rect_t* bbox, char** out_state_name) {
    if (!frame || !frame->data || !bbox || !out_state_name) return TK_ERROR_INVALID_ARGUMENT;
    int32_t d = 0;
    tk_error_code_t rc = classify_one(frame, bbox, nullptr, &d);
    if (rc != TK_SUCCESS) return rc;
    *out_state_name = strdup(d ? "closed" : "open");
    return *out_state_name ? TK_SUCCESS : TK_ERROR_OUT_OF_MEMORY;
}

/* ------------------------------------------------------------------ vision pipeline (object detection only) ------ */

static const char* const TK_COCO_LABELS[80] = {
    "person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light", "fire hydrant", "stop sign",
    "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow", "elephant", "bear", "zebra", "giraffe", "backpack", "umbrella",
    "handbag", "tie", "suitcase", "frisbee", "skis", "snowboard", "sports ball", "kite", "baseball bat", "baseball glove", "skateboard",
    "surfboard", "tennis racket", "bottle", "wine glass", "cup", "fork", "knife", "spoon", "bowl", "banana", "apple", "sandwich", "orange",
    "broccoli", "carrot", "hot dog", "pizza", "donut", "cake", "chair", "couch", "potted plant", "bed", "dining table", "toilet", "tv", "laptop",
    "mouse", "remote", "keyboard", "cell phone", "microwave", "oven", "toaster", "sink", "refrigerator", "book", "clock", "vase", "scissors",
    "teddy bear", "hair drier", "toothbrush"};

struct tk_vision_pipeline_s {
    tk_object_detector_t* detector = nullptr;
    tk_depth_estimator_t* depth = nullptr;
    TkFusion fusion; /* this pipeline's distance trackers (the reference keeps one process-wide set) */
    std::mutex mu;
    bool detection_enabled = false, depth_enabled = false;
    uint32_t max_objects = 0;
    float fx = 0.0f, fy = 0.0f;
};

tk_error_code_t tk_vision_pipeline_create(tk_vision_pipeline_t** out_pipeline, const tk_vision_pipeline_config_t* config) {
    if (!out_pipeline || !config) return TK_ERROR_INVALID_ARGUMENT;
    if (!config->object_detection_model_path) return vfail(TK_ERROR_INVALID_ARGUMENT, "object_detection_model_path missing");
    std::unique_ptr<tk_vision_pipeline_s> p(new tk_vision_pipeline_s());
    /* the reference's detector configuration (tk_vision_pipeline.c:368-379): 640x640, the 80 COCO labels, IoU 0.5 */
    tk_object_detector_config_t dc{};
    dc.backend = config->backend;
    dc.gpu_device_id = config->gpu_device_id;
    dc.model_path = config->object_detection_model_path;
    dc.input_width = 640;
    dc.input_height = 640;
    dc.class_labels = (const char**)TK_COCO_LABELS;
    dc.class_count = 80;
    dc.confidence_threshold = config->object_confidence_threshold;
    dc.iou_threshold = 0.5f;
    /* a detector that fails to load disables the analysis, it does not fail the pipeline (tk_vision_pipeline.c:380-385) */
    if (tk_object_detector_create(&p->detector, &dc) != TK_SUCCESS) p->detector = nullptr;
    p->detection_enabled = p->detector != nullptr;
    p->max_objects = config->max_detected_objects;
    p->fx = config->focal_length_x;
    p->fy = config->focal_length_y;
    /* the depth estimator at the reference's fixed 256 x 256 (tk_vision_pipeline.c:388-401); a missing or unloadable model disables the
     * analysis, it does not fail the pipeline */
    if (config->depth_estimation_model_path && config->depth_estimation_model_path->path_str) {
        tk_depth_estimator_config_t zc{};
        zc.backend = config->backend;
        zc.gpu_device_id = config->gpu_device_id;
        zc.model_path = config->depth_estimation_model_path;
        zc.input_width = 256;
        zc.input_height = 256;
        if (tk_depth_estimator_create(&p->depth, &zc) != TK_SUCCESS) p->depth = nullptr;
    }
    p->depth_enabled = p->depth != nullptr;
    *out_pipeline = p.release();
    return TK_SUCCESS;
}

void tk_vision_pipeline_destroy(tk_vision_pipeline_t** pipeline) {
    if (!pipeline || !*pipeline) return;
    if ((*pipeline)->detector) tk_object_detector_destroy(&(*pipeline)->detector);
    if ((*pipeline)->depth) tk_depth_estimator_destroy(&(*pipeline)->depth);
    delete *pipeline;
    *pipeline = nullptr;
}

tk_error_code_t tk_vision_pipeline_update_config(tk_vision_pipeline_t* pipeline, const tk_vision_runtime_config_t* config) {
    if (!pipeline || !config) return TK_ERROR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(pipeline->mu);
    pipeline->detection_enabled = config->enable_object_detection;
    pipeline->depth_enabled = config->enable_depth_estimation;
    if (pipeline->detector) tk_object_detector_update_thresholds(pipeline->detector, config->object_confidence_threshold, config->iou_threshold);
    return TK_SUCCESS;
}

void tk_vision_result_destroy(tk_vision_result_t** result) {
    if (!result || !*result) return;
    tk_vision_result_t* r = *result;
    for (size_t i = 0; i < r->object_count; ++i) {
        free((void*)r->objects[i].label);
        free(r->objects[i].recognized_text);
        free(r->objects[i].attributes);
    }
    free(r->objects);
    if (r->depth_map) tk_depth_estimator_free_map(&r->depth_map); /* tk_vision_pipeline.c:321-326 */
    free(r);
    *result = nullptr;
}

tk_error_code_t tk_vision_pipeline_process_frame(tk_vision_pipeline_t* pipeline, const tk_video_frame_t* video_frame,
                                                 tk_vision_analysis_flags_t analysis_flags, const tk_rect_t* ocr_roi, uint64_t timestamp_ns,
                                                 tk_vision_result_t** out_result) {
    (void)ocr_roi;
    if (!pipeline || !video_frame || !out_result) return TK_ERROR_INVALID_ARGUMENT;
    tk_vision_result_t* result = (tk_vision_result_t*)calloc(1, sizeof(tk_vision_result_t));
    if (!result) return TK_ERROR_OUT_OF_MEMORY;
    result->source_frame_timestamp_ns = timestamp_ns;
    bool enabled, depth_enabled;
    {
        std::lock_guard<std::mutex> lk(pipeline->mu);
        enabled = pipeline->detection_enabled;
        depth_enabled = pipeline->depth_enabled;
    }
    if (enabled && (analysis_flags & TK_VISION_ANALYZE_OBJECT_DETECTION) && pipeline->detector) {
        /* perform_object_detection (tk_vision_pipeline.c:435-494): detections -> objects, then the per-box attributes; a failure is
         * logged and the frame goes on without this analysis (:190-197) */
        tk_detection_result_t* det = nullptr;
        size_t n = 0;
        std::vector<int32_t> color, door;
        bool have_attr = false;
        tk_object_detector_s* d = pipeline->detector;
        bool ok = video_frame->data != nullptr;
        if (ok && !d->own) {
            /* one job of the shared engine: the network on this frame (batched with the other handles' frames) and, while the frame is still on
             * the device, the attributes of the boxes that will be reported; RGBA / padded frames have no attribute pass (the reference reads
             * them as packed RGB8 and classifies garbage) */
            DetReq r;
            const uint32_t bpp = (uint32_t)bytes_per_pixel(video_frame);
            r.frame = video_frame->data; r.w = video_frame->width; r.h = video_frame->height; r.stride = video_frame->stride ? video_frame->stride : video_frame->width * bpp;
            r.bpp = bpp; r.conf = d->conf; r.iou = d->iou; r.fast = d->fast; r.want_attr = bpp == 3 && r.stride == r.w * 3; r.max_objects = pipeline->max_objects;
            ok = r.w >= 2 && r.h >= 2 && r.stride >= r.w * bpp;
            if (ok) {
                d->sh->submit(&r);
                ok = r.ok;
            }
            if (ok) {
                d->last.assign(1, std::vector<TkDetection>());
                d->last[0].swap(r.dets);
                ok = to_results(d, d->last[0], video_frame, &det, &n) == TK_SUCCESS;
                have_attr = r.have_attr;
                color.swap(r.color);
                door.swap(r.door);
            }
            if (ok && pipeline->max_objects && n > pipeline->max_objects) n = pipeline->max_objects; /* results are score-descending */
        } else if (ok) {
            ok = tk_object_detector_detect(d, video_frame, &det, &n) == TK_SUCCESS;
            if (ok && pipeline->max_objects && n > pipeline->max_objects) n = pipeline->max_objects;
            if (ok && n > 0) {
                std::vector<int32_t> rects(4 * n);
                color.assign(n, -1);
                door.assign(n, 0);
                for (size_t i = 0; i < n; ++i) { rects[4 * i] = det[i].bbox.x; rects[4 * i + 1] = det[i].bbox.y; rects[4 * i + 2] = det[i].bbox.w; rects[4 * i + 3] = det[i].bbox.h; }
                have_attr = d->own->classify_boxes(0, (int)n, rects.data(), color.data(), door.data());
            }
        }
        if (ok && n > 0) {
            result->objects = (tk_vision_object_t*)calloc(n, sizeof(tk_vision_object_t));
            ok = result->objects != nullptr;
        }
        if (have_attr && (color.size() < n || door.size() < n)) have_attr = false;
        if (ok) {
            for (size_t i = 0; i < n; ++i) {
                tk_vision_object_t& o = result->objects[i];
                o.class_id = det[i].class_id;
                o.label = det[i].label ? strdup(det[i].label) : nullptr;
                o.confidence = det[i].confidence;
                o.bbox = det[i].bbox;
                if (have_attr) {
                    std::string a = std::string("color:") + TK_COLOR_NAMES[color[i]];
                    if (det[i].label && strstr(det[i].label, "door")) a += std::string(",state:") + (door[i] ? "closed" : "open");
                    o.attributes = strdup(a.c_str());
                }
            }
            result->object_count = n;
            result->valid_analyses_mask |= TK_VISION_RESULT_OBJECT_DETECTION;
        }
        if (det) tk_object_detector_free_results(&det);
    }
    /* depth estimation (tk_vision_pipeline.c:200-209, perform_depth_estimation :496-507): the map is owned by the result */
    if (depth_enabled && (analysis_flags & TK_VISION_ANALYZE_DEPTH_ESTIMATION) && pipeline->depth && video_frame->data) {
        if (tk_depth_estimator_estimate(pipeline->depth, video_frame, &result->depth_map) == TK_SUCCESS) result->valid_analyses_mask |= TK_VISION_RESULT_DEPTH_ESTIMATION;
    }
    /* object / depth fusion (:250-256, fuse_object_depth :653-713): a distance and a metric size per object */
    const bool can_fuse = (result->valid_analyses_mask & TK_VISION_RESULT_OBJECT_DETECTION) && (result->valid_analyses_mask & TK_VISION_RESULT_DEPTH_ESTIMATION);
    if ((analysis_flags & TK_VISION_ANALYZE_FUSION_DISTANCE) && can_fuse) {
        if (result->object_count > 0) {
            std::vector<TkBox> boxes(result->object_count);
            std::vector<uint32_t> cls(result->object_count);
            for (size_t i = 0; i < result->object_count; ++i) {
                const tk_rect_t& b = result->objects[i].bbox;
                boxes[i] = {b.x, b.y, b.w, b.h};
                cls[i] = result->objects[i].class_id;
            }
            std::vector<TkFused> fused;
            {
                std::lock_guard<std::mutex> lk(pipeline->mu);
                pipeline->fusion.fuse(boxes.data(), cls.data(), boxes.size(), result->depth_map->data, result->depth_map->width, result->depth_map->height,
                                      video_frame->width, video_frame->height, pipeline->fx, pipeline->fy, &fused);
            }
            for (size_t i = 0; i < result->object_count; ++i) {
                if (!fused[i].valid) continue; /* no valid depth under the box: the fields stay 0 */
                result->objects[i].distance_meters = fused[i].distance_m;
                result->objects[i].width_meters = fused[i].width_m;
                result->objects[i].height_meters = fused[i].height_m;
                result->objects[i].is_partially_occluded = false;
            }
        }
        result->valid_analyses_mask |= TK_VISION_RESULT_FUSION_DISTANCE;
    }
    *out_result = result;
    return TK_SUCCESS;
}

} /* extern "C" */
