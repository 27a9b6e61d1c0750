/*
 * tk_onnx_graph.h — an ONNX file as a graph: nodes with attributes, initialisers, declared inputs / outputs.  Wire-format reader only
 * (onnx.proto3 field numbers), no ONNX Runtime:
 *   ModelProto.graph = 7
 *   GraphProto.node = 1, .initializer = 5, .input = 11, .output = 12
 *   NodeProto.input = 1, .output = 2, .name = 3, .op_type = 4, .attribute = 5
 *   AttributeProto.name = 1, .f = 2, .i = 3, .s = 4, .t = 5, .g = 6 (a sub-graph: the branches of If), .floats = 7, .ints = 8, .type = 20
 *   TensorProto.dims = 1, .data_type = 2 (1 FLOAT, 6 INT32, 7 INT64, 9 BOOL, 10 FLOAT16), .float_data = 4, .int32_data = 5, .int64_data = 7,
 *               .name = 8, .raw_data = 9
 *   ValueInfoProto.name = 1, .type = 2 -> TypeProto.tensor_type = 1 -> {elem_type = 1, shape = 2 -> dim = 1 -> {dim_value = 1, dim_param = 2}}
 * The reference hands silero_vad.onnx (src/sensors/tk_vad_silero.c:110-280) and the MiDaS depth model (src/vision/tk_depth_midas.c:231-283)
 * to ONNX Runtime; csrc/nn/tk_onnx_exec runs such graphs on the GPU from this description.  Host-only code, tested without a GPU.
 */
#ifndef TK_ONNX_GRAPH_H
#define TK_ONNX_GRAPH_H

#include <stdint.h>

#include <map>
#include <memory>
#include <string>
#include <vector>

struct TkOnnxGraph;

struct TkOnnxTensor {
    std::vector<int64_t> dims;
    int dtype = 0;              /* ONNX data type of the file */
    std::vector<float> f;       /* FLOAT / FLOAT16 payloads as f32 */
    std::vector<int64_t> i;     /* INT32 / INT64 payloads */
    int64_t count() const { int64_t n = 1; for (int64_t d : dims) n *= d; return n; }
};

struct TkOnnxAttr {
    float f = 0.0f;
    int64_t i = 0;
    std::string s;
    std::vector<int64_t> ints;
    std::vector<float> floats;
    TkOnnxTensor t;
    bool has_t = false;
    std::shared_ptr<TkOnnxGraph> g; /* a sub-graph: then_branch / else_branch of an If node, the body of a Loop / Scan (nesting depth limited by the reader) */
};

struct TkOnnxNode {
    std::string op, name;
    std::vector<std::string> in, out;
    std::map<std::string, TkOnnxAttr> attr;
    int64_t ai(const char* k, int64_t dflt) const { auto it = attr.find(k); return it == attr.end() ? dflt : it->second.i; }
    float af(const char* k, float dflt) const { auto it = attr.find(k); return it == attr.end() ? dflt : it->second.f; }
    bool has(const char* k) const { return attr.find(k) != attr.end(); }
    const std::vector<float>* afloats(const char* k) const { auto it = attr.find(k); return it == attr.end() ? nullptr : &it->second.floats; }
    const std::vector<int64_t>* aints(const char* k) const { auto it = attr.find(k); return it == attr.end() ? nullptr : &it->second.ints; }
    std::string as(const char* k, const char* dflt) const { auto it = attr.find(k); return it == attr.end() ? std::string(dflt) : it->second.s; }
};

struct TkOnnxValueInfo {
    std::string name;
    int elem_type = 0;
    std::vector<int64_t> dims; /* -1 for symbolic / unknown dimensions */
};

struct TkOnnxGraph {
    std::vector<TkOnnxNode> nodes; /* file order = a valid execution order (ONNX requires topological order) */
    std::map<std::string, TkOnnxTensor> init;
    std::vector<TkOnnxValueInfo> inputs, outputs; /* inputs exclude initialisers */
    std::string error;
    bool load(const char* path);
    /* every node of the graph and of its sub-graphs, outer ones first (op checks, Constant collection) */
    void all_nodes(std::vector<const TkOnnxNode*>* out) const;
};

#endif
