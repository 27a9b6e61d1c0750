/*
 * tk_onnx_exec.h — an ONNX graph executed node by node on the GPU, without ONNX Runtime.
 *
 * The reference creates an ORT session per model file and calls Run() on it: the Silero VAD (src/sensors/tk_vad_silero.c:110-280) and the
 * MiDaS depth network (src/vision/tk_depth_midas.c:231-283, 397-440).  Here the graph description comes from tk_onnx_graph (wire-format
 * reader) and every node becomes one or a few HIP kernels on the owner's stream.  Two graph classes are covered:
 *   1-D signal graphs (VAD): Conv over [1, C, L], Pad, Slice, Pow, LSTM, ReduceMean ...
 *   2-D convolutional image graphs (MiDaS v2.1 class: EfficientNet-lite / ResNeXt encoders + feature-fusion decoder): Conv over
 *   [N, C, H, W] with groups (dense ones as im2col + the exact fp32 MFMA GEMM of tk_nn_kernels, grouped / depthwise ones as a direct
 *   kernel), Relu / Clip / LeakyRelu / Sigmoid / HardSigmoid / HardSwish, Add / Mul / ... with numpy broadcasting, Concat, Resize / Upsample
 *   (nearest, linear; half_pixel, pytorch_half_pixel, align_corners, asymmetric), MaxPool, AveragePool, GlobalAveragePool,
 *   BatchNormalization, MatMul, Gemm, Softmax, and the layout-only ops.
 *   token-sequence graphs (DPT / Swin-transformer depth models, the class the reference names: src/vision/tk_depth_midas.c:8,
 *   tests/tk_cortex_test.cpp:42): LayerNormalization, Erf / Gelu, batched MatMul, Gather, ReduceSum / ReduceL2 / ReduceMax / ReduceMin, Expand,
 *   Max / Min / Where, ConvTranspose, strided Slice, Shape and integer shape arithmetic, rank-6 Transpose / Reshape (tk_onnx_exec_seq.hip).
 *   control flow: If whose condition is host data (Equal / Less / Greater / Not / And / Or on integer or bool tensors — the per-sample-rate
 *   switch of Silero-class VAD exports): the chosen branch's nodes run in the same value map, its initialisers and Constant nodes are
 *   resident like the outer graph's.  Loop (trip count and / or condition as host data: a constant, or integer arithmetic on the iteration
 *   number inside the body; loop-carried values, scan outputs stacked along a new leading axis) and Scan (state values + scan inputs sliced
 *   along axis 0, forward or reversed; scan outputs along axis 0) run their body graph in the same value map, one fresh set of arena tensors
 *   per iteration — what exporters emit for recurrences written as Python loops.  A condition computed from activations (Equal / Less / Greater
 *   on float tensors of up to 64 elements) is evaluated on the host after a stream synchronisation.
 * Anything else fails at load time with the op's name.
 * Arithmetic: fp32; every contraction is one k-ascending fma chain per output element (input channel outer, kernel row, kernel column
 * inner; the bias enters last), the exact-math exp / tanh / sigmoid / sqrt of common/tk_exact_math.h.
 */
#ifndef TK_ONNX_EXEC_H
#define TK_ONNX_EXEC_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "tk_onnx_graph.h"

class TkOnnxExec {
public:
    struct Val {
        float* d = nullptr;            /* device data (float tensors) */
        std::vector<int64_t> shape;
        std::vector<int64_t> ints;     /* host data (int tensors: shapes, axes, slice bounds) */
        std::vector<float> host;       /* host copy of small float constants (scalars the host needs: Clip bounds, Resize scales) */
        bool is_int = false;
        int64_t count() const { int64_t n = 1; for (int64_t s : shape) n *= s; return n; }
    };
    std::string error;
    ~TkOnnxExec() { unload(); }
    void unload(); /* waits for the stream, frees constants and the arena (call before destroying the stream) */
    /* reads the file, checks every op against the supported list, uploads initialisers and Constant nodes.  The stream belongs to the
     * caller and must outlive this object; arena_floats bounds the activations of one run. */
    bool load(const char* path, int device, hipStream_t stream, size_t arena_floats);
    static bool ops_supported(const TkOnnxGraph& g, std::string* err);
    const TkOnnxGraph& graph() const { return g_; }
    /* one run: begin() forgets the previous run's activations, bind() names the inputs, run() launches every node in file order;
     * value() then finds any tensor by name (device memory inside the arena, valid until the next begin()) */
    void begin();
    void bind(const std::string& name, const Val& v) { vals_[name] = v; }
    bool run();
    const Val* value(const std::string& name) const { auto it = vals_.find(name); return it == vals_.end() ? nullptr : &it->second; }
    size_t arena_high_water() const { return arena_peak_; }

private:
    bool exec(const TkOnnxNode& nd, std::map<std::string, Val>& v);
    bool exec_image_op(const TkOnnxNode& nd, std::map<std::string, Val>& v, bool* handled);
    bool exec_seq_op(const TkOnnxNode& nd, std::map<std::string, Val>& v, bool* handled); /* tk_onnx_exec_seq.hip: the transformer (DPT / Swin) ops */
    float* alloc(int64_t n);
    bool add_const(const std::string& name, const TkOnnxTensor& t);
    TkOnnxGraph g_;
    int device_ = 0;
    hipStream_t stream_ = nullptr;
    std::map<std::string, Val> consts_;  /* initialisers + Constant nodes, resident */
    std::map<std::string, Val> vals_;
    std::map<std::string, float*> packed_; /* per Conv node: [M][K + 1] weights with the bias as the last column (dense convolutions) */
    std::vector<std::vector<int32_t>> staging_; /* host sources of this run's small uploads (Gather indices, masks): alive until the next begin() */
    float* arena_ = nullptr;
    size_t arena_cap_ = 0, arena_used_ = 0, arena_peak_ = 0;
};

#endif
