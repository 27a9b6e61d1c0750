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
    det.set_thresholds(0.9999, 0.5)
    assert det.detect(fr[0]) == []                                            # empty result set


def test_detector_errors(gpu):
    with pytest.raises(gpu.TkError) as e:
        gpu.ObjectDetector(backend=0)
    assert e.value.code == 4005
    with pytest.raises(gpu.TkError):
        gpu.ObjectDetector(width=100)
    with pytest.raises(gpu.TkError):
        gpu.ObjectDetector(model="/nonexistent/weights.tkyolo")
