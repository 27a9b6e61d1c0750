/* tk_vad_graph.hip — see tk_vad_graph.h: the recurrent-state and windowing logic around the generic graph executor */
#include "tk_vad_graph.h"

#include <stdio.h>
#include <string.h>

#define VQ(expr)                                                                                                  \
    do {                                                                                                          \
        hipError_t e__ = (expr);                                                                                  \
        if (e__ != hipSuccess) { error = std::string(#expr) + " failed: " + hipGetErrorString(e__); return false; } \
    } while (0)

TkVadGraph::~TkVadGraph() {
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    for (auto& s : states_) if (s.buf) (void)hipFree(s.buf);
    if (x_dev_) (void)hipFree(x_dev_);
    if (p_dev_) (void)hipFree(p_dev_);
    exec_.unload();
    if (stream_) (void)hipStreamDestroy(stream_);
}

bool TkVadGraph::check_supported(const TkOnnxGraph& g, std::string* err) {
    if (!TkOnnxExec::ops_supported(g, err)) return false;
    bool audio = false;
    for (const auto& vi : g.inputs) audio = audio || vi.elem_type == 1 || vi.elem_type == 0;
    if (!audio) { *err = "the graph has no float input for the audio window"; return false; }
    return true;
}

bool TkVadGraph::load(const char* path, int device, int window, int sample_rate) {
    device_ = device; window_ = window; sample_rate_ = sample_rate;
    VQ(hipSetDevice(device_));
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess) VQ(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, hi));
    else VQ(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    /* 16 MiB of activations: far more than a 30 ms window needs in any graph of this class */
    if (!exec_.load(path, device_, stream_, (size_t)1 << 22)) { error = exec_.error; return false; }
    const TkOnnxGraph& g = exec_.graph();
    if (!check_supported(g, &error)) return false;
    /* inputs: the first float input is the audio window; an integer input is the sample rate; the other float inputs are recurrent state */
    std::vector<const TkOnnxValueInfo*> extra;
    for (const auto& vi : g.inputs) {
        if (vi.elem_type == 7 || vi.elem_type == 6) { sr_in_ = vi.name; continue; }
        if (audio_in_.empty()) { audio_in_ = vi.name; continue; }
        extra.push_back(&vi);
    }
    if (audio_in_.empty()) { error = "the graph has no float input for the audio window"; return false; }
    if (g.outputs.empty()) { error = "the graph declares no outputs"; return false; }
    prob_out_ = g.outputs[0].name;
    for (size_t i = 0; i < extra.size(); ++i) {
        State st;
        st.in = extra[i]->name;
        const std::string cand[2] = {st.in + "n", st.in + "N"}; /* h -> hn, c -> cn, state -> stateN */
        for (const auto& o : g.outputs)
            if (o.name == cand[0] || o.name == cand[1]) st.out = o.name;
        if (st.out.empty() && i + 1 < g.outputs.size()) st.out = g.outputs[i + 1].name;
        if (st.out.empty()) { error = "recurrent input '" + st.in + "' has no matching output"; return false; }
        int64_t n = 1;
        for (int64_t d : extra[i]->dims) { const int64_t dd = d > 0 ? d : 1; st.shape.push_back(dd); n *= dd; } /* symbolic batch = 1 */
        if (st.shape.empty() || n > (1 << 20)) { error = "recurrent input '" + st.in + "' needs a static shape"; return false; }
        VQ(hipMalloc((void**)&st.buf, (size_t)n * 4));
        VQ(hipMemset(st.buf, 0, (size_t)n * 4));
        states_.push_back(st);
    }
    VQ(hipMalloc((void**)&p_dev_, 4));
    /* one dry run finds shape or broadcasting errors at load time, as session creation does in the reference */
    std::vector<float> zero((size_t)window_, 0.0f);
    float p = 0.0f;
    if (!infer(zero.data(), 1, &p)) return false;
    return reset();
}

bool TkVadGraph::reset() {
    VQ(hipSetDevice(device_));
    for (auto& s : states_) {
        int64_t n = 1;
        for (int64_t d : s.shape) n *= d;
        VQ(hipMemsetAsync(s.buf, 0, (size_t)n * 4, stream_));
    }
    VQ(hipStreamSynchronize(stream_));
    return true;
}

bool TkVadGraph::infer(const float* windows_host, int n, float* prob_host) {
    if (n <= 0) return true;
    VQ(hipSetDevice(device_));
    if (n > xcap_) {
        if (x_dev_) (void)hipFree(x_dev_);
        if (p_dev_) (void)hipFree(p_dev_);
        x_dev_ = p_dev_ = nullptr;
        const int want = n < 128 ? 128 : n;
        VQ(hipMalloc((void**)&x_dev_, (size_t)want * window_ * 4));
        VQ(hipMalloc((void**)&p_dev_, (size_t)want * 4));
        xcap_ = want;
    }
    VQ(hipMemcpyAsync(x_dev_, windows_host, (size_t)n * window_ * 4, hipMemcpyHostToDevice, stream_));
    for (int i = 0; i < n; ++i)
        if (!run_window(x_dev_ + (size_t)i * window_, p_dev_ + i)) return false;
    VQ(hipGetLastError());
    VQ(hipMemcpyAsync(prob_host, p_dev_, (size_t)n * 4, hipMemcpyDeviceToHost, stream_));
    VQ(hipStreamSynchronize(stream_));
    return true;
}

bool TkVadGraph::run_window(const float* x_dev, float* prob_dev) {
    exec_.begin();
    TkOnnxExec::Val in;
    in.d = const_cast<float*>(x_dev);
    in.shape = {1, window_};
    exec_.bind(audio_in_, in);
    if (!sr_in_.empty()) { TkOnnxExec::Val sr; sr.is_int = true; sr.shape = {1}; sr.ints = {sample_rate_}; exec_.bind(sr_in_, sr); }
    for (auto& s : states_) { TkOnnxExec::Val sv; sv.d = s.buf; sv.shape = s.shape; exec_.bind(s.in, sv); }
    if (!exec_.run()) { error = exec_.error; return false; }
    const TkOnnxExec::Val* po = exec_.value(prob_out_);
    if (!po || !po->d || po->count() < 1) { error = "the graph did not produce its first output"; return false; }
    VQ(hipMemcpyAsync(prob_dev, po->d, 4, hipMemcpyDeviceToDevice, stream_));
    for (auto& s : states_) { /* the next window starts from this window's final state */
        const TkOnnxExec::Val* so = exec_.value(s.out);
        int64_t n = 1;
        for (int64_t d : s.shape) n *= d;
        if (!so || !so->d || so->count() != n) { error = "state output '" + s.out + "' is missing or has the wrong size"; return false; }
        if (so->d != s.buf) VQ(hipMemcpyAsync(s.buf, so->d, (size_t)n * 4, hipMemcpyDeviceToDevice, stream_));
    }
    return true;
}
