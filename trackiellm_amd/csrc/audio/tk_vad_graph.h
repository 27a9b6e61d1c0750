/*
 * tk_vad_graph.h — a voice-activity-detection ONNX graph (the Silero VAD class: STFT as a strided Conv with a fixed basis, magnitude,
 * Conv + ReLU encoder, LSTM, 1x1 Conv + Sigmoid head, mean) executed on the GPU without ONNX Runtime.
 *
 * The reference creates an ORT session on silero_vad.onnx and runs it once per 30 ms window (src/sensors/tk_vad_silero.c:110-190 create,
 * :193-280 run_vad_inference: one float input [1, window], the first output's first element is the speech probability).  Here the graph
 * description comes from csrc/nn/tk_onnx_graph (wire-format reader) and every node becomes one small HIP kernel on the VAD stream
 * (csrc/nn/tk_onnx_exec, the executor shared with the depth network); the ops a VAD graph of this class uses:
 *   Conv (1-D, group 1), Relu, Sigmoid, Tanh, Sqrt, Abs, Neg, Exp, Log, Add, Sub, Mul, Div, Pow (numpy broadcasting), Slice, Concat, Pad
 *   (constant / reflect), Transpose, ReduceMean, LSTM (forward, one direction), and the layout-only ops Unsqueeze, Squeeze, Reshape, Flatten,
 *   Identity, Cast(float), Constant.
 * Anything else fails at load time with the op's name.  Recurrent inputs (h / c -> hn / cn, or state -> stateN; otherwise the i-th extra
 * float input pairs with the (i + 1)-th output) are kept on the device between windows and cleared by reset(); an int64 `sr` input is fed
 * the sample rate.  Arithmetic: fp32, k-ordered fma chains, the exact-math exp / tanh / sigmoid / sqrt of common/tk_exact_math.h.
 * No silero_vad.onnx exists offline: parity is pinned to a torch implementation of a generated graph of this class
 * (tests/golden/vad_graph.npz), not to the published checkpoint.
 */
#ifndef TK_VAD_GRAPH_H
#define TK_VAD_GRAPH_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../nn/tk_onnx_exec.h"

class TkVadGraph {
public:
    std::string error;
    ~TkVadGraph();
    bool load(const char* path, int device, int window, int sample_rate);
    /* host-only: every node's op is one this path runs, the graph has outputs and a float input */
    static bool check_supported(const TkOnnxGraph& g, std::string* err);
    /* windows: [n][window] float on the host, processed in order (recurrent state carried from one to the next) -> probabilities[n] */
    bool infer(const float* windows_host, int n, float* prob_host);
    bool reset();
    bool stateful() const { return !states_.empty(); }
    int node_count() const { return (int)exec_.graph().nodes.size(); }

private:
    struct State { std::string in, out; float* buf = nullptr; std::vector<int64_t> shape; };
    bool run_window(const float* x_dev, float* prob_dev);
    TkOnnxExec exec_; /* the node-by-node executor (csrc/nn/tk_onnx_exec.h) */
    int device_ = 0, window_ = 480, sample_rate_ = 16000;
    hipStream_t stream_ = nullptr;
    std::vector<State> states_;
    std::string audio_in_, sr_in_, prob_out_;
    float *x_dev_ = nullptr, *p_dev_ = nullptr;
    int xcap_ = 0;
};

#endif
