"""Depth path without a GPU: the numpy oracle (oracle/depth_oracle.py) pinned to torch (tests/golden/depth_net.npz) and to the reference's
own fusion unit test; the host-only parts of the product (fusion C-ABI, ONNX probe) against the oracle."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import depth_oracle as DO  # noqa: E402
import onnx_util as OX  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "depth_net.npz")


def test_oracle_network_matches_torch_fixture():
    g = np.load(GOLD)
    W = OX.depth_weights(int(g["seed"]))
    v = DO.run_graph(OX.depth_spec(), OX.depth_consts(W), {"input": g["input"]})
    scale = float(np.abs(g["output"]).max())
    assert v["output"].shape == g["output"].shape
    assert np.abs(v["output"] - g["output"]).max() <= 2e-5 * scale  # fp32 summation order differs; torch is the pin
    for k in ("b2", "up1"):  # intermediate taps (stored as f16): the squeeze-excite branch and the align_corners upsampling
        t = g["tap_" + k].astype(np.float32)
        assert np.abs(v[k] - t).max() <= 2e-3 * float(np.abs(t).max())


def test_oracle_resize_modes_known_answers():
    x = np.arange(4, dtype=np.float32).reshape(1, 1, 2, 2)  # [[0, 1], [2, 3]]
    # align_corners x2 on 2 -> 4 samples at 0, 1/3, 2/3, 1
    y = DO.resize(x, 4, 4, np.float32(2), np.float32(2), "linear", "align_corners")
    assert np.allclose(y[0, 0, 0], [0, 1 / 3, 2 / 3, 1], atol=1e-6) and np.allclose(y[0, 0, :, 0], [0, 2 / 3, 4 / 3, 2], atol=1e-6)
    # half_pixel x2: coordinates -0.25, 0.25, 0.75, 1.25 clamped to [0, 1]
    y = DO.resize(x, 4, 4, np.float32(2), np.float32(2), "linear", "half_pixel")
    assert np.allclose(y[0, 0, 0], [0, 0.25, 0.75, 1], atol=1e-6)
    # nearest, asymmetric + floor (the Upsample op): each sample repeated
    y = DO.resize(x, 4, 4, np.float32(2), np.float32(2), "nearest", "asymmetric", "floor")
    assert np.array_equal(y[0, 0], [[0, 0, 1, 1], [0, 0, 1, 1], [2, 2, 3, 3], [2, 2, 3, 3]])


def test_oracle_metric_conversion_known_answers():
    """src/vision/tk_depth_midas.c:471-499: inverse depth, min -> 10 m, max -> 0.1 m, flat map -> 10 m"""
    raw = np.array([[0.0, 1.0], [2.0, 4.0]], np.float32)
    m = DO.to_metric(raw)
    assert m[0, 0] == np.float32(10.0) and m[1, 1] == np.float32(10.0) - np.float32(1.0) * np.float32(np.float32(10.0) - np.float32(0.1))
    assert m[0, 1] == np.float32(10.0) - np.float32(0.25) * np.float32(9.9)
    assert np.all(DO.to_metric(np.full((3, 3), 2.5, np.float32)) == np.float32(10.0))


def test_oracle_fusion_replays_reference_unit_test():
    """src/vision/src/object_analysis.rs:286-347 (test_kalman_filter_smoothes_distance): a 20x20 box over depth 10, then the box moved by one
    pixel over depth 12: first distance within 0.1 of 10, second strictly between 10 and 12 (one Kalman step: K = 1.1 / 1.6)"""
    f = DO.Fusion()
    d1 = np.zeros((100, 100), np.float32)
    d1[10:30, 10:30] = 10.0
    r1 = f.fuse([(10, 10, 20, 20)], [1], d1, 100, 100, 300.0, 300.0)
    assert len(r1) == 1 and abs(float(r1[0][0]) - 10.0) < 0.1
    d2 = np.zeros((100, 100), np.float32)
    d2[11:31, 11:31] = 12.0
    r2 = f.fuse([(11, 11, 20, 20)], [1], d2, 100, 100, 300.0, 300.0)
    assert len(r2) == 1 and 10.0 < float(r2[0][0]) < 12.0
    assert float(r2[0][0]) == pytest.approx(10.0 + (1.1 / 1.6) * 2.0, abs=1e-5)
    assert float(r2[0][1]) == pytest.approx(20 * float(r2[0][0]) / 300.0, rel=1e-6)


def _scene(rng, n_boxes, dw, dh, fw, fh):
    depth = (0.5 + 9.0 * rng.random((dh, dw))).astype(np.float32)
    depth[rng.random((dh, dw)) < 0.1] = 0.0  # invalid pixels
    boxes = []
    for _ in range(n_boxes):
        w, h = int(rng.integers(2, fw // 2)), int(rng.integers(2, fh // 2))
        boxes.append((int(rng.integers(0, fw - w)), int(rng.integers(0, fh - h)), w, h))
    return depth, boxes


def test_product_fusion_matches_oracle(tk):
    """the exported fusion C-ABI (host code, no GPU): raw distances, tracker matching over several frames, ageing — equal to the oracle to
    the last bit (same float32 operations in the same order)"""
    rng = np.random.default_rng(3)
    tk.fusion_reset()
    f = DO.Fusion()
    fw, fh, dw, dh = 640, 480, 64, 48
    base = None
    for frame in range(9):
        depth, boxes = _scene(rng, 6, dw, dh, fw, fh)
        if base is None:
            base = boxes
        # boxes drift slowly so that trackers match; every third frame two of them vanish (ageing), one tiny box has < 10 valid depths
        boxes = [(b[0] + frame, b[1] + frame // 2, b[2], b[3]) for b in base]
        if frame % 3 == 2:
            boxes = boxes[:4]
        boxes.append((5, 5, 8, 8))
        classes = list(range(len(boxes)))
        for b in boxes:
            assert tk.fusion_raw_distance(b, depth, fw, fh) == DO.raw_distance(b, depth, fw, fh)
        want = [w for w in f.fuse(boxes, classes, depth, fw, fh, 500.0, 400.0) if w is not None]
        got = tk.fuse_data(boxes, classes, depth, fw, fh, 500.0, 400.0)
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert (g[3], g[4], g[5]) == (w[0], w[1], w[2]) and g[1] == 1.0
    tk.fusion_reset()


def test_fusion_null_and_degenerate_inputs(tk):
    tk.fusion_reset()
    depth = np.full((10, 10), 3.0, np.float32)
    assert tk.fusion_raw_distance((0, 0, 0, 0), depth, 10, 10) == np.float32(-1)          # empty box
    assert tk.fusion_raw_distance((0, 0, 2, 2), depth, 10, 10) == np.float32(-1)          # 3 x 3 = 9 < 10 valid depths
    assert tk.fusion_raw_distance((0, 0, 5, 5), depth, 10, 10) == np.float32(3.0)
    assert tk.fusion_raw_distance((8, 8, 50, 50), depth, 10, 10) == np.float32(-1)        # leaves the map
    assert tk.fuse_data([], [], depth, 10, 10, 1.0, 1.0) == []


def test_depth_onnx_probe(tk, tmp_path):
    W = OX.depth_weights(11)
    p = tmp_path / "depth.onnx"
    p.write_bytes(OX.depth_model(W, 64, 64))
    assert tk.depth_onnx_probe(str(p)) == len(OX.depth_spec())
    q = tmp_path / "depth_einsum.onnx"
    q.write_bytes(OX.depth_model(W, 64, 64, extra_op="Einsum"))  # an op of the DPT / Swin exports: refused with its name
    with pytest.raises(tk.TkError) as e:
        tk.depth_onnx_probe(str(q))
    assert e.value.code == 4001 and "Einsum" in str(e.value)
    with pytest.raises(tk.TkError):
        tk.depth_onnx_probe(str(tmp_path / "missing.onnx"))


def test_swin_class_graph_oracle_matches_torch_fixture_and_is_accepted(tk, tmp_path):
    """the DPT / Swin-transformer graph class (the depth model the reference names: src/vision/tk_depth_midas.c:8, tests/tk_cortex_test.cpp:42):
    the numpy restatement of its ops agrees with the torch-made fixture (tests/golden/depth_swin.npz), and the op check of the GPU executor
    accepts the file (26 distinct ops, among them LayerNormalization, Erf, Gelu, Gather, ReduceL2, Where on a bool mask, ConvTranspose,
    strided Slice, Shape-driven Reshape, rank-6 Transpose)"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "depth_swin.npz"))
    W = OX.swin_weights(int(g["seed"]))
    v = DO.run_graph(OX.swin_spec(), OX.swin_consts(W), {"input": g["input"]})
    scale = float(np.abs(g["output"]).max())
    assert np.abs(v["output"] - g["output"]).max() <= 2e-5 * scale
    for k in ("b2_x2", "x2g", "u1"):
        assert np.abs(v[k] - g["tap_" + k]).max() <= 2e-5 * float(np.abs(g["tap_" + k]).max()), k
    ops = {n["op"] for n in OX.swin_spec()}
    assert {"LayerNormalization", "Erf", "Gelu", "Gather", "ReduceL2", "ReduceSum", "Where", "ConvTranspose", "Shape", "Softmax", "MatMul"} <= ops
    p = tmp_path / "swin.onnx"
    p.write_bytes(OX.swin_model(W))
    assert tk.depth_onnx_probe(str(p)) == len(OX.swin_spec())
    q = tmp_path / "swin_bad.onnx"
    q.write_bytes(OX.swin_model(W, extra_op="NonMaxSuppression"))
    with pytest.raises(tk.TkError) as e:
        tk.depth_onnx_probe(str(q))
    assert "NonMaxSuppression" in str(e.value)
    # a graph that declares no outputs is refused at load
    r = tmp_path / "noout.onnx"
    r.write_bytes(OX.model([OX.node("Relu", ["input"], ["y"])], [], [OX.value_info("input", 1, [1, 3, 8, 8])], []))
    with pytest.raises(tk.TkError) as e:
        tk.depth_onnx_probe(str(r))
    assert "no outputs" in str(e.value)
