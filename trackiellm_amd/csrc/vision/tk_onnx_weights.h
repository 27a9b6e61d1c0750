/*
 * tk_onnx_weights.h — convolution weights out of an ONNX file, without ONNX Runtime.
 *
 * The reference hands the detector's .onnx to ONNX Runtime (src/vision/tk_object_detector.c:93-152).  This path runs its own
 * YOLOv8n graph (common/tk_yolov8n_graph.h), so all it needs from the file are the Conv nodes' weight / bias initialisers, in
 * execution order — an Ultralytics export lists them in exactly the order the graph here walks its 63 convolutions (Conv+BN
 * already fused), followed by the constant DFL 1x1 conv, which is skipped (the DFL expectation is computed in k_yolo_decode).
 *
 * Only the protobuf wire format is parsed (onnx.proto3 field numbers):
 *   ModelProto.graph = 7;  GraphProto.node = 1, .initializer = 5
 *   NodeProto.input = 1, .op_type = 4
 *   TensorProto.dims = 1, .data_type = 2 (1 = FLOAT, 10 = FLOAT16), .float_data = 4, .name = 8, .raw_data = 9
 * Host-only code; parsing is tested without a GPU.
 */
#ifndef TK_ONNX_WEIGHTS_H
#define TK_ONNX_WEIGHTS_H

#include <stdint.h>

#include <string>
#include <vector>

struct TkOnnxConv {
    int cout = 0, cin = 0, kh = 0, kw = 0;
    std::vector<float> w; /* [cout][cin][kh][kw] as stored */
    std::vector<float> b; /* [cout], zeros when the node has no bias */
};

class TkOnnxWeights {
public:
    std::vector<TkOnnxConv> convs; /* Conv nodes in file (= execution) order */
    std::string error;
    static bool looks_like_onnx(const char* path); /* cheap sniff: a length-delimited field 7 is reachable at top level */
    bool load(const char* path);
};

#endif
