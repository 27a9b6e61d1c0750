"""CPU: detector-stream oracle against its pins (compiled reference pre-processor, torch YOLOv8n fixture)."""
import hashlib
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
from make_vision_golden import frames  # noqa: E402  (seeded frame generator only; no reference code)


def test_preprocess_restatement_matches_compiled_reference_fixture():
    g = np.load(os.path.join(GOLD, "preprocess_small.npz"))
    fr = frames()
    assert np.array_equal(O.preprocess(fr["rand_96x64"], 64, 64).view(np.uint32), g["small_a"].view(np.uint32))
    assert np.array_equal(O.preprocess(fr["rand_37x23"], 32, 32).view(np.uint32), g["small_b"].view(np.uint32))
    for name in ("rand_640x480", "rand_640x640", "gray128_640x480"):
        y = O.preprocess(fr[name], 640, 640)
        assert hashlib.sha256(y.tobytes()).digest() == g["sha_" + name].tobytes(), name


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_preprocess_against_live_compiled_reference():
    rng = np.random.default_rng(9)
    for (w, h, tw, th) in ((50, 40, 32, 32), (640, 480, 640, 640), (33, 17, 64, 96)):
        f = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(O.preprocess(f, tw, th).view(np.uint32), O.ref_preprocess(f, tw, th).view(np.uint32))


def test_preprocess_is_not_identity_and_honours_stride():
    f = np.arange(8 * 8 * 3, dtype=np.uint8).reshape(8, 8, 3)
    y = O.preprocess(f, 8, 8, mean=np.zeros(3), std=np.ones(3))
    assert not np.allclose(y[0, 1, 1] * 255.0, f[1, 1, 0])          # ratio 7/8 resamples every pixel
    pad = np.zeros((8, 40), np.uint8)
    pad[:, :24] = f.reshape(8, 24)
    out = np.empty((3, 8, 8), np.float32)
    O.lib().orc_preprocess(O.ptr(pad), 8, 8, 40, 3, O.ptr(out), 8, 8, O.ptr(np.zeros(3, np.float32)), O.ptr(np.ones(3, np.float32)), 0)
    assert np.array_equal(out, y)


def test_yolo_oracle_matches_torch_fixture():
    g = np.load(os.path.join(GOLD, "yolo_tiny.npz"))
    orc = O.OracleYolo(nc=80, seed=5, cls_bias=-4.0)
    raw = orc.forward(g["x"])
    assert np.array_equal(raw, g["oracle_raw"])
    assert np.abs(raw - g["torch_raw"]).max() < 2e-4 * max(1.0, np.abs(g["torch_raw"]).max())
    assert len(orc.layers()) == 63 and sum(l["w"].size + l["b"].size for l in orc.layers()) > 3_000_000


def test_yolo_oracle_matches_independent_torch_graph_at_640x640():
    """the BASELINE geometry (640 x 640 -> 8400 anchors x 144 channels), not only the 64 x 64 toy: 2048 sampled head-map values and per-scale
    statistics of an INDEPENDENT torch implementation of the published YOLOv8n graph (tests/golden/make_vision_golden.py: make_yolo_full) —
    the graph walker the oracle shares with the product is pinned by a second implementation at the size the bench runs"""
    g = np.load(os.path.join(GOLD, "yolo_full_640.npz"))
    x = np.random.default_rng(11).standard_normal((1, 640, 640, 3)).astype(np.float32)  # make_vision_golden.yolo_full_input()
    raw = O.OracleYolo(nc=80, seed=5, cls_bias=-4.0).forward(x)[0]
    assert raw.shape == (8400, 144)
    scale = float(g["scale"])
    got = raw[g["idx"][:, 0], g["idx"][:, 1]]
    assert np.abs(got - g["torch_vals"]).max() < 2e-4 * max(1.0, scale)
    for (a, b), st in zip(((0, 6400), (6400, 8000), (8000, 8400)), g["stats"]):
        assert abs(raw[a:b].mean() - st[0]) < 1e-4 and abs(np.abs(raw[a:b]).max() - st[1]) < 2e-4 * scale and abs(raw[a:b].std() - st[2]) < 1e-4


def test_yolo_post_matches_independent_torch_detections():
    """DFL decode + class-aware NMS against an independent torch implementation (Ultralytics decode, torchvision-style batched NMS):
    anchor indices and classes identical, boxes / scores within 1e-3 (SURVEY 8c's yolo_tiny_dets)."""
    g = np.load(os.path.join(GOLD, "yolo_tiny_dets.npz"))
    orc = O.OracleYolo(nc=80, seed=5, cls_bias=-1.0)
    raw = orc.forward(g["x"])[0]
    boxes, cls, anc = orc.post(raw, 160, 160, float(g["conf"]), float(g["iou"]))
    assert int(g["n_candidates"]) - len(g["torch_anchors"]) >= 3          # the fixture does exercise suppression
    assert np.array_equal(anc, g["torch_anchors"]) and np.array_equal(cls, g["torch_cls"])
    assert np.abs(boxes - g["torch_boxes"]).max() < 1e-3


@pytest.mark.skipif(not O.have_ref() or not os.path.exists(os.path.join(O.ROOT, "oracle", "_ref", "libtkref_attr.so")), reason="compiled reference not built")
def test_attribute_vectors_fixture_is_what_the_compiled_reference_answers():
    import json
    from make_vision_golden import attribute_vector_frames
    j = json.load(open(os.path.join(GOLD, "attribute_vectors.json")))
    for name, (frame, box) in attribute_vector_frames().items():
        color, door = O.ref_attributes(frame, box)
        assert j[name]["compiled_reference"] == {"color": color, "door": door}


def test_nms_properties():
    orc = O.OracleYolo(nc=80, seed=5, cls_bias=-2.0)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((1, 64, 64, 3)).astype(np.float32)
    raw = orc.forward(x)[0]
    boxes, cls, anc = orc.post(raw, 64, 64, 0.05, 0.5)
    assert len(boxes) > 3
    assert np.all(np.diff(boxes[:, 4]) <= 0)                              # score-descending
    assert len(set(anc.tolist())) == len(anc)                             # no duplicates
    # idempotence: survivors never suppress each other
    for i in range(len(boxes)):
        for j in range(i + 1, len(boxes)):
            if cls[i] == cls[j]:
                iw = min(boxes[i, 2], boxes[j, 2]) - max(boxes[i, 0], boxes[j, 0])
                ih = min(boxes[i, 3], boxes[j, 3]) - max(boxes[i, 1], boxes[j, 1])
                if iw > 0 and ih > 0:
                    inter = iw * ih
                    u = (boxes[i, 2] - boxes[i, 0]) * (boxes[i, 3] - boxes[i, 1]) + (boxes[j, 2] - boxes[j, 0]) * (boxes[j, 3] - boxes[j, 1]) - inter
                    assert inter / u <= 0.5 + 1e-6
    # IoU threshold 1.0 keeps every candidate; threshold 0 keeps at most one per overlapping class cluster
    allb, _, _ = orc.post(raw, 64, 64, 0.05, 1.0)
    assert len(allb) >= len(boxes)
    none, _, _ = orc.post(raw, 64, 64, 0.999999, 0.5)
    assert len(none) == 0                                                  # empty result is legal
