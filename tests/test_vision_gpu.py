"""GPU parity: pre-processor, fp32 MFMA GEMM, YOLOv8n head maps and NMS indices against the oracle."""
import hashlib
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
from make_vision_golden import frames  # noqa: E402


def test_preprocess_bit_exact_vs_compiled_reference_outputs(gpu):
    g = np.load(os.path.join(GOLD, "preprocess_small.npz"))
    fr = frames()
    assert np.array_equal(gpu.preprocess(fr["rand_96x64"], 64, 64).view(np.uint32), g["small_a"].view(np.uint32))
    assert np.array_equal(gpu.preprocess(fr["rand_37x23"], 32, 32).view(np.uint32), g["small_b"].view(np.uint32))
    for name in ("rand_640x480", "rand_640x640", "gray128_640x480"):   # BASELINE sizes, incl. the reference test's gray frame
        y = gpu.preprocess(fr[name], 640, 640)
        assert hashlib.sha256(y.tobytes()).digest() == g["sha_" + name].tobytes(), name
        assert np.array_equal(y.view(np.uint32), O.preprocess(fr[name], 640, 640).view(np.uint32))


def test_preprocess_stride_rgba_and_errors(gpu):
    rng = np.random.default_rng(4)
    f = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    pad = np.zeros((20, 128), np.uint8)
    pad[:, :90] = f.reshape(20, 90)
    assert np.array_equal(gpu.preprocess(pad, 32, 32, stride=128, width=30), gpu.preprocess(f, 32, 32))
    rgba = np.concatenate([f, np.full((20, 30, 1), 255, np.uint8)], 2)
    assert np.array_equal(gpu.preprocess(rgba, 32, 32, rgba=True), gpu.preprocess(f, 32, 32))
    import ctypes as C
    assert gpu.lib().tk_preprocessor_resize_and_normalize_to_chw(None, None, 1, 1, None, None) == 1001


def test_detector_head_maps_bit_exact(gpu):
    det = gpu.ObjectDetector(width=64, height=64, conf=0.05, max_batch=2)
    orc = O.OracleYolo(nc=80, seed=5, cls_bias=-4.0)
    g = np.load(os.path.join(GOLD, "yolo_tiny.npz"))
    rng = np.random.default_rng(21)
    x = np.concatenate([g["x"], rng.standard_normal((1, 64, 64, 3)).astype(np.float32)])
    raw = det.forward_raw(x)
    want = orc.forward(x)
    assert np.array_equal(raw, want), np.abs(raw - want).max()
    assert np.abs(raw[0] - g["torch_raw"][0]).max() < 2e-4 * max(1.0, np.abs(g["torch_raw"]).max())
    for b in range(2):
        boxes, cls, anc = det.last_boxes(b)
        wb, wc, wa = orc.post(want[b], 64, 64, 0.05, 0.5)
        assert np.array_equal(anc, wa) and np.array_equal(cls, wc)          # detection INDICES identical
        assert np.array_equal(boxes, wb)


def test_detector_full_size_frame_bit_exact(gpu):
    """BASELINE geometry: one 640x640 network input (8400 anchors): head maps and NMS survivors equal the oracle's, bit for bit"""
    det = gpu.ObjectDetector(width=640, height=640, conf=0.05)
    orc = O.OracleYolo(nc=80, seed=5, cls_bias=-4.0)
    x = np.random.default_rng(33).standard_normal((1, 640, 640, 3)).astype(np.float32)
    raw = det.forward_raw(x)
    want = orc.forward(x)
    assert raw.shape == (1, 8400, 144)
    assert np.array_equal(raw, want), np.abs(raw - want).max()
    boxes, cls, anc = det.last_boxes(0)
    wb, wc, wa = orc.post(want[0], 640, 640, 0.05, 0.5)
    assert np.array_equal(anc, wa) and np.array_equal(cls, wc) and np.array_equal(boxes, wb)


def test_detector_full_size_frame_against_the_independent_torch_graph(gpu):
    """the HIP detector itself against the independent torch YOLOv8n at 640 x 640 (yolo_full_640.npz: sampled head-map values), 2e-4 of the
    head maps' scale — north_star's fp tolerance on boxes, checked without the oracle in between"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "yolo_full_640.npz"))
    det = gpu.ObjectDetector(width=640, height=640, conf=0.05)
    x = np.random.default_rng(11).standard_normal((1, 640, 640, 3)).astype(np.float32)
    raw = det.forward_raw(x)[0]
    got = raw[g["idx"][:, 0], g["idx"][:, 1]]
    assert np.abs(got - g["torch_vals"]).max() < 2e-4 * max(1.0, float(g["scale"]))


def test_detector_weights_from_onnx_file(gpu, tmp_path):
    """an .onnx detector file (Conv initialisers in execution order + DFL conv, raw and packed float storage) goes in through
    tk_object_detector_create's model_path: head maps equal the oracle that owns the same weights"""
    import onnx_util as X
    orc = O.OracleYolo(nc=80, seed=11, cls_bias=-3.0)
    path = tmp_path / "yolov8n.onnx"
    path.write_bytes(X.yolo_model(orc.layers(), raw=False))
    det = gpu.ObjectDetector(model=str(path), width=64, height=64, conf=0.05)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 64, 64, 3)).astype(np.float32)
    raw = det.forward_raw(x)
    want = orc.forward(x)
    assert np.array_equal(raw, want), np.abs(raw - want).max()
    # f16 storage: weights are the f16-rounded ones (biases stay f32): compare with a detector fed through the flat container path
    h16 = [dict(L, w=L["w"].astype(np.float16).astype(np.float32)) for L in orc.layers()]
    path.write_bytes(X.yolo_model(orc.layers(), f16=True))
    det16 = gpu.ObjectDetector(model=str(path), width=64, height=64, conf=0.05)
    flat = tmp_path / "flat.tkyolo"
    with open(flat, "wb") as f:
        f.write(b"TKYOLO1\0" + np.array([len(h16), 80], np.int32).tobytes())
        for L in h16:
            f.write(np.array([L["cin"], L["cout"], L["k"], L["s"]], np.int32).tobytes() + L["w"].tobytes() + L["b"].tobytes())
    detf = gpu.ObjectDetector(model=str(flat), width=64, height=64, conf=0.05)
    assert np.array_equal(det16.forward_raw(x), detf.forward_raw(x))


def test_detector_detections_match_independent_torch_fixture(gpu):
    """HIP decode / rank / NMS kernels against detections computed by an independent torch post-processor (yolo_tiny_dets.npz):
    anchor indices and classes identical, boxes within 1e-3 px."""
    g = np.load(os.path.join(GOLD, "yolo_tiny_dets.npz"))
    det = gpu.ObjectDetector(model="synthetic://yolov8n?seed=5&cls_bias=-1", width=160, height=160, conf=float(g["conf"]), iou=float(g["iou"]))
    det.forward_raw(g["x"])
    boxes, cls, anc = det.last_boxes()
    assert np.array_equal(anc, g["torch_anchors"]) and np.array_equal(cls, g["torch_cls"])
    assert np.abs(boxes - g["torch_boxes"]).max() < 1e-3
    det.close()


def test_detector_full_frame_path_and_batch(gpu):
    """u8 frame -> preprocess -> network -> NMS -> original-frame rects, 160x160 network input, 2 frames"""
    rng = np.random.default_rng(8)
    fr = [rng.integers(0, 256, (120, 200, 3), dtype=np.uint8) for _ in range(2)]
    det = gpu.ObjectDetector(model="synthetic://yolov8n?seed=5&cls_bias=-1", width=160, height=160, conf=0.3, iou=0.5, max_batch=2)
    orc = O.OracleYolo(nc=80, seed=5, cls_bias=-1.0)
    res = det.detect_batch(fr)
    one = det.detect(fr[1])
    assert one == res[1]
    for b in range(2):
        x = O.preprocess(fr[b], 160, 160, nhwc=True)[None]
        raw = orc.forward(x)[0]
        wb, wc, wa = orc.post(raw, 160, 160, 0.3, 0.5)
        assert len(res[b]) == len(wb) and len(wb) > 0
        sx, sy = np.float32(200) / np.float32(160), np.float32(120) / np.float32(160)
        for (cid, label, confd, (x0, y0, w, h)), bb, cc in zip(res[b], wb, wc):
            assert cid == cc and label == gpu.COCO80[cc].encode() and np.float32(confd) == bb[4]
            assert (x0, y0, w, h) == (int(bb[0] * sx), int(bb[1] * sy), int((bb[2] - bb[0]) * sx), int((bb[3] - bb[1]) * sy))
    # stride == 0 means tightly packed rows, as the reference's callers leave it (tk_object_detector.c:235 assumes width * 3)
    import ctypes as C
    from trackiellm_amd.vision import VideoFrame, DetectionResult
    f0 = VideoFrame(200, 120, 0, 0, fr[1].ctypes.data)
    rp, n = C.POINTER(DetectionResult)(), C.c_size_t(0)
    assert gpu.lib().tk_object_detector_detect(det.h, C.byref(f0), C.byref(rp), C.byref(n)) == 0
    assert [(rp[i].class_id, rp[i].bbox.x, rp[i].bbox.y, rp[i].bbox.w, rp[i].bbox.h) for i in range(n.value)] == \
           [(c, r[0], r[1], r[2], r[3]) for c, _, _, r in one]
    gpu.lib().tk_object_detector_free_results(C.byref(rp))
    det.set_thresholds(0.9999, 0.5)
    assert det.detect(fr[0]) == []                                            # empty result set


@pytest.mark.skipif(not O.have_ref() or not os.path.exists(os.path.join(O.ROOT, "oracle", "_ref", "libtkref_attr.so")), reason="compiled reference not built")
def test_box_attributes_equal_the_compiled_reference(gpu):
    """tk_classify_dominant_color / tk_classify_door_state on the GPU against the reference's own C compiled from its sources:
    random frames and boxes (also partly outside the frame for the colour histogram), saturated primaries, grays, a striped door"""
    rng = np.random.default_rng(2)
    frames = [rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)]
    prim = np.zeros((40, 72, 3), np.uint8)
    for i, c in enumerate([(255, 0, 0), (255, 255, 0), (0, 255, 0), (0, 255, 255), (0, 0, 255), (255, 0, 255), (0, 0, 0), (255, 255, 255), (128, 128, 128)]):
        prim[:, 8 * i:8 * i + 8] = c
    frames.append(prim)
    stripes = np.full((60, 60, 3), 30, np.uint8)
    stripes[0::4] = 220
    stripes[1::4] = 220                                   # two bright rows, two dark rows: the rows above and below always differ by 190
    frames.append(stripes)
    frames.append((rng.integers(0, 4, (50, 50, 1)) * 60 + rng.integers(0, 20, (50, 50, 3))).astype(np.uint8))  # low saturation
    for f in frames:
        H, W = f.shape[:2]
        boxes = [(0, 0, W, H), (3, 2, 9, 7), (W // 2, H // 2, W // 2, H // 2), (1, 1, W - 2, H - 2)]
        boxes += [tuple(int(v) for v in (rng.integers(1, W - 8), rng.integers(1, H - 8), rng.integers(3, 8), rng.integers(3, 8))) for _ in range(6)]
        for b in boxes:
            assert gpu.classify_attributes(f, b) == O.ref_attributes(f, b), (f.shape, b)
        b = (W - 6, H - 6, 12, 12)   # partly outside: colour only (the reference's door loop has no bounds check; its colour loop skips)
        big = np.zeros((H + 16, W, 3), np.uint8)
        big[:H] = f                  # the reference's door loop reads rows past the box: give it memory to read
        assert gpu.classify_attributes(f, b)[0] == O.ref_attributes(big[:H], b)[0]
    for i, name in enumerate(["red", "yellow", "green", "cyan", "blue", "magenta", "black", "white", "gray"]):
        assert gpu.classify_attributes(prim, (8 * i, 0, 8, 40))[0] == name
    assert gpu.classify_attributes(stripes, (5, 5, 40, 40))[1] == "closed"
    assert gpu.classify_attributes(prim, (0, 0, 8, 40))[1] == "open"


def test_reference_attribute_test_vectors(gpu):
    """tests/tk_attribute_classifier_test.c replayed verbatim: the pure-red 100 x 100 frame with box (10, 10, 80, 80) is "red" (:21-54);
    the door vector (:56-91) gets what the reference's IMPLEMENTATION answers — "open", not the "closed" its test asserts: the frame has
    vertical bands only and tk_classify_door_state thresholds the difference between the rows above and below (ABI_NOTES.md).  Expected
    values: tests/golden/attribute_vectors.json = outputs of the compiled reference."""
    import json
    from make_vision_golden import attribute_vector_frames
    j = json.load(open(os.path.join(GOLD, "attribute_vectors.json")))
    fr = attribute_vector_frames()
    red, box = fr["red_frame"]
    assert list(box) == j["red_frame"]["bbox"] and red.shape == (100, 100, 3) and (red[..., 0] == 255).all() and not red[..., 1:].any()
    got = gpu.classify_attributes(red, box)
    assert got[0] == j["red_frame"]["reference_test_expects"]["color"] == "red"
    assert {"color": got[0], "door": got[1]} == j["red_frame"]["compiled_reference"]
    door, box = fr["door_frame"]
    got = gpu.classify_attributes(door, box)
    assert {"color": got[0], "door": got[1]} == j["door_frame"]["compiled_reference"]
    assert got[1] == "open" and j["door_frame"]["reference_test_expects"]["door"] == "closed"   # the recorded contradiction


def test_vision_pipeline_object_detection(gpu):
    """tk_vision_pipeline_*: detections copied into tk_vision_object_t with the per-box colour attribute of the resident frame"""
    rng = np.random.default_rng(9)
    frame = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    pipe = gpu.VisionPipeline(conf=0.5, max_objects=12)
    det = gpu.ObjectDetector(model="synthetic://yolov8n?seed=5&cls_bias=-0.45", conf=0.5)
    ts, mask, objs = pipe.process(frame, flags=1 | 2 | 4, timestamp_ns=77)
    want = det.detect(frame)[:12]
    assert ts == 77 and mask == 1                      # depth / OCR requested, never reported
    assert [(o[0], o[1], o[3]) for o in objs] == [(w[0], w[1], w[3]) for w in want]
    assert len(objs) > 0
    for cls, label, conf, bbox, attr in objs:
        assert attr == ("color:" + gpu.classify_attributes(frame, bbox)[0]).encode()
    assert pipe.process(frame, flags=2)[1:] == (0, [])   # object detection not requested
    pipe.update(0.5, 0.5, enable=False)
    assert pipe.process(frame, flags=1)[1:] == (0, [])
    pipe.update(0.999, 0.5, enable=True)
    assert pipe.process(frame, flags=1)[1] == 1
    pipe.close()


def test_detector_errors(gpu):
    with pytest.raises(gpu.TkError) as e:
        gpu.ObjectDetector(backend=0)
    assert e.value.code == 4005
    with pytest.raises(gpu.TkError):
        gpu.ObjectDetector(width=100)
    with pytest.raises(gpu.TkError):
        gpu.ObjectDetector(model="/nonexistent/weights.tkyolo")


def test_detectors_on_one_file_share_an_engine_and_give_the_solo_results(gpu):
    """round 6: tk_object_detector_t / tk_vision_pipeline_t handles opened on the same model file share the weights and ONE
    batched engine; their one-frame calls (the reference's API shape, src/vision/tk_object_detector.c:182-219) are coalesced by a scheduler.
    Six handles driven at once from six threads each return exactly what a lone handle returns for that frame — detections, boxes and the
    pipeline's per-box attributes — and the engine's counters show that frames really travelled together."""
    import threading
    rng = np.random.default_rng(77)
    K = 6
    frames = [rng.integers(0, 256, (480, 640, 3), dtype=np.uint8) for _ in range(K)]
    model = "synthetic://yolov8n?seed=5&cls_bias=-0.40"
    solo_det, solo_pipe = [], []
    for f in frames:                                   # one handle alive at a time: nothing to coalesce with
        d = gpu.ObjectDetector(model=model, conf=0.5)
        assert d.share_stats()[0] == 1
        solo_det.append(d.detect(f))
        d.close()
        p = gpu.VisionPipeline(model=model, conf=0.5, max_objects=20)
        solo_pipe.append(p.process(f, flags=1)[2])
        p.close()
    assert sum(len(r) for r in solo_det) > K and len({len(r) for r in solo_det}) > 1
    dets = [gpu.ObjectDetector(model=model, conf=0.5) for _ in range(K // 2)]
    pipes = [gpu.VisionPipeline(model=model, conf=0.5, max_objects=20) for _ in range(K - K // 2)]
    assert dets[0].share_stats()[0] == K               # detector handles and pipelines' detectors: one engine
    other = gpu.ObjectDetector(model="synthetic://yolov8n?seed=6&cls_bias=-0.40", conf=0.5)
    assert other.share_stats()[0] == 1                 # another file, another engine
    got = [None] * K
    bar = threading.Barrier(K)

    def run(i):
        bar.wait()
        for rep in range(3):                           # the same frame thrice: every round coalesces anew
            got[i] = dets[i].detect(frames[i]) if i < len(dets) else pipes[i - len(dets)].process(frames[i], flags=1)[2]

    th = [threading.Thread(target=run, args=(i,)) for i in range(K)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(K):
        assert got[i] == (solo_det[i] if i < len(dets) else solo_pipe[i]), "handle %d differs from its solo result" % i
    handles, batches, nframes, widest = dets[0].share_stats()
    assert nframes == 3 * K and batches < nframes and widest >= 2, (handles, batches, nframes, widest)
    # thresholds are per handle: a stricter handle does not change its neighbours' results
    dets[0].set_thresholds(0.9, 0.5)
    strict = dets[0].detect(frames[0])
    assert len(strict) < len(solo_det[0]) and dets[1].detect(frames[1]) == solo_det[1]
    for h in dets + pipes + [other]:
        h.close()


def test_detector_runs_a_yolov5u_class_onnx_through_the_graph_executor(gpu, tmp_path):
    """the reference names yolov5nu.onnx (/root/reference/src/cortex/tk_cortex_main.h:71, tests/tk_cortex_test.cpp:41,
    src/vision/tk_object_detector.c:93-152 hands any file to ONNX Runtime).  A file that is not the 63-convolution YOLOv8n topology — here a
    seeded YOLOv5u-class graph (C3 blocks, SPPF, PAN, anchor-free DFL head; 262 nodes spelled as an Ultralytics export) — loads through
    tk_object_detector_create and runs node by node on the library's ONNX executor; its [1, 4 + nc, anchors] output equals the INDEPENDENT torch
    modules' (tests/golden/make_vision_golden.py: torch_yolo5) within 2e-5 of its scale, and the detections — decode + NMS on the device — have
    the torch post-processor's indices."""
    import onnx_util as X
    d = np.load(os.path.join(GOLD, "yolo5nu_tiny.npz"))
    nc, H = 80, d["x"].shape[2]
    Wt = X.yolo5_weights(int(d["seed"]), nc, float(d["cls_bias"]))
    path = tmp_path / "yolov5nu.onnx"
    path.write_bytes(X.yolo5_model(Wt, nc, H, H))
    conf, iou = float(d["conf"]), float(d["iou"])
    det = gpu.ObjectDetector(model=str(path), width=H, height=H, conf=conf, iou=iou)
    assert det.is_graph()
    out = det.forward_graph(d["x"])
    scale = float(np.abs(d["out"]).max())
    assert out.shape == (1,) + d["out"].shape and np.abs(out[0] - d["out"]).max() <= 2e-5 * scale, np.abs(out[0] - d["out"]).max() / scale
    boxes, cls, anc = det.last_boxes()
    assert np.array_equal(anc, d["torch_anchors"]) and np.array_equal(cls, d["torch_cls"]) and len(anc) < int(d["n_candidates"])
    assert np.abs(boxes - d["torch_boxes"]).max() <= 1e-3 * scale
    ob, oc, oa = O.yolo_post_out(out[0], nc, conf, iou)                     # and the oracle's decode + NMS on the product's own output: bit for bit
    assert np.array_equal(anc, oa) and np.array_equal(cls, oc) and np.array_equal(boxes, ob)
    with pytest.raises(gpu.TkError):
        det.forward_raw(np.zeros((1, H, H, 3), np.float32))                  # raw head maps are the hard-wired path's
    # the reference's call: a frame through tk_object_detector_detect = pre-process (planar for a file's graph) + graph + decode + NMS
    rng = np.random.default_rng(4)
    frame = rng.integers(0, 256, (96, 160, 3), dtype=np.uint8)
    got = det.detect(frame)
    chw = gpu.preprocess(frame, H, H)
    det.forward_graph(chw[None])
    b2, c2, a2 = det.last_boxes()
    sx, sy = 160 / H, 96 / H
    want = [(int(c), (int(b[0] * np.float32(sx)), int(b[1] * np.float32(sy)), int((b[2] - b[0]) * np.float32(sx)), int((b[3] - b[1]) * np.float32(sy)))) for b, c in zip(b2, c2)]
    assert [(g[0], g[3]) for g in got] == want and len(got) > 0
    # two frames in one call: the graph runs frame by frame, results per frame
    frame2 = rng.integers(0, 256, (96, 160, 3), dtype=np.uint8)
    both = det.detect_batch([frame, frame2])
    assert both[0] == got and both[1] == det.detect(frame2)
    det.close()
    # a YOLOv8n-topology file keeps the hard-wired path
    orc = O.OracleYolo(nc=80, seed=11, cls_bias=-3.0)
    p8 = tmp_path / "yolov8n.onnx"
    p8.write_bytes(X.yolo_model(orc.layers()))
    d8 = gpu.ObjectDetector(model=str(p8), width=64, height=64)
    assert not d8.is_graph()
    d8.close()
    # a graph the executor cannot run is refused at create with the op's name; so is an output of another geometry at the first frame
    bad = tmp_path / "bad.onnx"
    bad.write_bytes(X.yolo5_model(Wt, nc, H, H, extra_op="NonMaxSuppression"))
    with pytest.raises(gpu.TkError) as e:
        gpu.ObjectDetector(model=str(bad), width=H, height=H)
    assert "NonMaxSuppression" in e.value.detail
    other = gpu.ObjectDetector(model=str(path), width=2 * H, height=2 * H)   # the file's anchor constants are for H x H
    with pytest.raises(gpu.TkError):
        other.detect(frame)
    other.close()


def test_detector_fast_contraction_gate(gpu):
    """the opt-in fast contraction (tk_mi355x_detector_set_fast_contraction: convolutions on the f16 matrix pipe with split operands) against the
    exact path and the independent torch fixtures — VERDICT r05 item 8's gate: (a) head maps at 640 x 640 within 1e-5 of their scale of the
    exact path and inside the torch fixture's 2e-4; (b) on yolo_tiny_dets.npz anchor indices and classes identical to the torch post-processor's,
    boxes within 1e-3 px; (c) on a full-size frame the same (anchor, class) set as the exact path, boxes within 1e-3 px, order equal wherever
    two scores differ by more than 1e-6; (d) switched off again the handle returns the exact path's bits."""
    g = np.load(os.path.join(GOLD, "yolo_full_640.npz"))
    det = gpu.ObjectDetector(width=640, height=640, conf=0.05)
    x = np.random.default_rng(11).standard_normal((1, 640, 640, 3)).astype(np.float32)
    exact = det.forward_raw(x)
    det.set_fast_contraction(True)
    fast = det.forward_raw(x)
    scale = float(np.abs(exact).max())
    assert not np.array_equal(fast, exact)                                    # the other kernels did run
    assert np.abs(fast - exact).max() < 1e-5 * scale, np.abs(fast - exact).max() / scale
    assert np.abs(fast[0][g["idx"][:, 0], g["idx"][:, 1]] - g["torch_vals"]).max() < 2e-4 * max(1.0, float(g["scale"]))
    det.set_fast_contraction(False)
    assert np.array_equal(det.forward_raw(x), exact)
    det.close()
    # (c) the bench's detector (dense detections: ~100 per frame at confidence 0.5)
    dense = gpu.ObjectDetector(model="synthetic://yolov8n?seed=5&cls_bias=-0.45", width=640, height=640, conf=0.5, iou=0.5)
    frame = np.random.default_rng(1).integers(0, 256, (640, 640, 3), dtype=np.uint8)   # the bench's kind of frame, through the whole path
    dense.detect(frame)
    eb, ec, ea = dense.last_boxes(0)
    dense.set_fast_contraction(True)
    dense.detect(frame)
    fb, fc, fa = dense.last_boxes(0)
    assert len(ea) > 20 and sorted(zip(ea.tolist(), ec.tolist())) == sorted(zip(fa.tolist(), fc.tolist()))
    by_anchor = {(int(a), int(c)): i for i, (a, c) in enumerate(zip(fa, fc))}
    for i, (a, c) in enumerate(zip(ea, ec)):
        j = by_anchor[(int(a), int(c))]
        assert np.abs(eb[i, :4] - fb[j, :4]).max() < 1e-3 and abs(float(eb[i, 4]) - float(fb[j, 4])) < 1e-6
        if i != j:                                                            # a swap only between detections the exact path scores within 1e-6
            assert abs(float(eb[i, 4]) - float(eb[j, 4])) < 1e-6
    dense.close()
    t = np.load(os.path.join(GOLD, "yolo_tiny_dets.npz"))
    tiny = gpu.ObjectDetector(model="synthetic://yolov8n?seed=5&cls_bias=-1", width=160, height=160, conf=float(t["conf"]), iou=float(t["iou"]))
    tiny.set_fast_contraction(True)
    tiny.forward_raw(t["x"])
    boxes, cls, anc = tiny.last_boxes()
    assert np.array_equal(anc, t["torch_anchors"]) and np.array_equal(cls, t["torch_cls"])
    assert np.abs(boxes - t["torch_boxes"]).max() < 1e-3
    tiny.close()


def test_handles_with_and_without_fast_contraction_share_an_engine(gpu):
    """handles of one model file may differ in the opt-in fast contraction: their one-frame calls ride separate jobs of the shared engine — the exact
    handles keep returning the exact path's bits, the fast ones what a fast handle returns alone — driven from four threads at once"""
    import threading
    rng = np.random.default_rng(21)
    K = 4
    fr = [rng.integers(0, 256, (160, 160, 3), dtype=np.uint8) for _ in range(K)]
    mk = lambda: gpu.ObjectDetector(model="synthetic://yolov8n?seed=5&cls_bias=-1", width=160, height=160, conf=0.3, iou=0.5)  # noqa: E731
    solo = []
    for i in range(K):
        d = mk()
        d.set_fast_contraction(i % 2 == 1)
        solo.append(d.detect(fr[i]))
        d.close()
    assert any(len(s) > 0 for s in solo)
    dets = [mk() for _ in range(K)]
    for i, d in enumerate(dets):
        d.set_fast_contraction(i % 2 == 1)
    got = [None] * K
    bar = threading.Barrier(K)

    def run(i):
        bar.wait()
        for _ in range(3):
            got[i] = dets[i].detect(fr[i])

    th = [threading.Thread(target=run, args=(i,)) for i in range(K)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert got == solo
    handles, batches, frames, widest = dets[0].share_stats()
    assert handles == K and frames == 3 * K and batches >= 2
    for d in dets:
        d.close()
