/*
 * tk_vision_engine.h — detector stream on the GPU: frame upload, the reference pre-processor's
 * arithmetic (src/vision/tk_image_preprocessor.c:43-69,156-160) as a HIP kernel, YOLOv8n on the
 * fp32 MFMA GEMM, DFL decode + on-device NMS.  Batched over B frames (one per concurrent cycle).
 */
#ifndef TK_VISION_ENGINE_H
#define TK_VISION_ENGINE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include <memory>

#include "../common/tk_ggml_blocks.h"
#include "../common/tk_yolo_post.h"
#include "../common/tk_yolov8n_graph.h"
#include "../nn/tk_onnx_exec.h"

struct TkPreprocessArgs {
    const uint8_t* src;  /* device, interleaved */
    uint32_t in_w, in_h, in_stride, bpp;
    float* dst;          /* device */
    uint32_t out_w, out_h;
    float mean[3], std_dev[3];
    int nhwc;            /* 0: planar CHW (reference layout), 1: NHWC (detector input) */
};
void tk_launch_preprocess(const TkPreprocessArgs& a, hipStream_t s);

struct TkDetection {
    float x1, y1, x2, y2, score;
    int32_t cls, anchor;
};

class TkYoloModel {
public:
    int device = 0, nc = TK_YOLO_NC;
    std::vector<TkConvSpec> specs;
    std::vector<float*> w, b; /* device: [cout][k*k*cin], [cout] */
    std::string error;
    ~TkYoloModel();
    bool init(int device, int nc);
    bool fill_synthetic(uint64_t seed, float cls_bias);
    bool set_layer(int idx, const float* w_host, const float* b_host);
    bool load_file(const char* path); /* "TKYOLO1\0" flat container: see INTEGRATION.md */
    /* Conv initialisers of an Ultralytics YOLOv8 ONNX export, in execution order (vision/tk_onnx_weights.h); the trailing DFL conv is skipped.
     * A file that is not the 63-convolution YOLOv8n topology — the reference names yolov5nu.onnx (src/cortex/tk_cortex_main.h:71,
     * tests/tk_cortex_test.cpp:41): C3 blocks, the same [1, 4 + nc, anchors] output — becomes a `generic` model: every engine runs the file's own
     * graph node by node (csrc/nn/tk_onnx_exec) and decodes its decoded-box output (k_yolo_decode_out) instead of the raw head maps. */
    bool load_onnx(const char* path);
    bool generic = false;    /* the graph executor path */
    std::string onnx_path;   /* generic: the file every engine loads */
    std::string graph_in, graph_out; /* generic: the graph's float input / first output */
    size_t param_count() const;
};

class TkDetector {
public:
    TkYoloModel* model = nullptr;
    int in_w = 640, in_h = 640, max_batch = 1;
    float conf = 0.5f, iou = 0.5f;
    /* opt-in: the convolutions contract on the f16 matrix pipe with split operands (TkGemm::fast) — ~1e-6 of scale off the exact chain */
    bool fast = false;
    float mean[3] = {0.485f, 0.456f, 0.406f}, std_dev[3] = {0.229f, 0.224f, 0.225f};
    std::string error;
    hipStream_t stream = nullptr;

    ~TkDetector();
    bool init(TkYoloModel* m, int in_w, int in_h, int max_batch);
    /* frames: B host frames of identical geometry; results per frame, score-descending */
    bool detect(int B, const uint8_t* const* frames, uint32_t w, uint32_t h, uint32_t stride, uint32_t bpp, std::vector<std::vector<TkDetection>>* out);
    /* test hooks: run the network on an already pre-processed NHWC tensor / fetch the raw head maps */
    bool forward_tensor(int B, const float* nhwc_host, std::vector<float>* raw_out /* [B][8400][64+nc] */);
    /* generic models: the graph on a pre-processed planar tensor [B][3][H][W]; out [B][4 + nc][anchors] = the file's own output */
    bool forward_graph(int B, const float* nchw_host, std::vector<float>* out);
    bool fetch(int B, std::vector<std::vector<TkDetection>>* out); /* download the NMS survivors of the last enqueue */
    /* per-box attributes on frame b of the last detect() batch, still resident on the device (reference: the CPU loops of
     * src/vision/tk_attribute_classifier.c run per detection, tk_vision_pipeline.c:462-485): rects[n] = {x, y, w, h} in frame pixels;
     * color[n] = dominant-colour bin 0..8, door_closed[n] = 1 when the strong-vertical-edge density exceeds 10 % */
    bool classify_boxes(int b, int n, const int32_t* rects, int32_t* color, int32_t* door_closed);

    bool enqueue(int B); /* pre-processed input already in `input`; network + decode + NMS on `stream` */
    float* input = nullptr;        /* [B][H][W][3] */
    int n_anchors = 0;

private:
    friend struct TkGpuOps;
    uint8_t* frame_dev = nullptr;
    size_t frame_cap = 0;
    uint32_t last_w = 0, last_h = 0, last_stride = 0; /* geometry of the frames of the last detect() */
    int last_B = 0;
    int32_t* attr_dev = nullptr; /* [cap][4] rects then [cap][2] results */
    int attr_cap = 0;
    bool run_attributes(const uint8_t* frame, uint32_t w, uint32_t h, int n, const int32_t* rects, int32_t* color, int32_t* door_closed);
    std::unique_ptr<TkOnnxExec> exec; /* generic models: this engine's own instance of the file's graph (constants resident, one arena per run) */
    bool run_graph(int B, float* host_out); /* generic: `input` holds B planar frames -> cand[b][anchor]; host_out (optional) [B][4 + nc][anchors] */
    float* arena = nullptr;
    size_t arena_floats = 0, arena_used = 0;
    float* col = nullptr;
    size_t col_floats = 0;
    TkT heads[3];
    tk_yolo_cand_t* cand = nullptr;     /* [B][n_anchors] decoded */
    int32_t* order = nullptr;           /* [B][MAX_CAND] anchor ids, sorted */
    int32_t* n_cand = nullptr;          /* [B] */
    uint64_t* mask = nullptr;           /* [B][MAX_CAND][MAX_CAND/64] */
    TkDetection* kept = nullptr;        /* [B][MAX_DET] */
    int32_t* n_kept = nullptr;          /* [B] */
};

/* the same classification on a host frame (tightly packed RGB8), temporary device buffers on `device`: the reference's stand-alone
 * tk_classify_dominant_color / tk_classify_door_state entry points */
bool tk_classify_boxes_host(int device, const uint8_t* frame, uint32_t w, uint32_t h, int n, const int32_t* rects, int32_t* color,
                            int32_t* door_closed, std::string* error);

#endif
