#!/usr/bin/env python3
"""One 640 x 640 frame through the detector's two paths (developer tool, needs an MI355X): the hard-wired YOLOv8n graph and a YOLOv5u-class .onnx of the
nano model's widths run node by node on the ONNX executor (tests/onnx_util.yolo5_model with YOLO5_CH = 16 .. 256) — ms per tk_object_detector_detect."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import onnx_util as X  # noqa: E402
import trackiellm_amd as tk  # noqa: E402

X.YOLO5_CH = (16, 32, 64, 128, 256)
rng = np.random.default_rng(3)
frame = rng.integers(0, 256, (640, 640, 3), dtype=np.uint8)
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "yolov5nu.onnx")
    W = X.yolo5_weights(5, 80, -1.5)
    open(path, "wb").write(X.yolo5_model(W, 80, 640, 640))
    print("graph file: %.1f MB, %d parameters" % (os.path.getsize(path) / 1e6, sum(v.size for v in W.values())))
    for name, model in (("hard-wired YOLOv8n", "synthetic://yolov8n?seed=5&cls_bias=-0.45"), ("YOLOv5u-class .onnx on the graph executor", path)):
        det = tk.ObjectDetector(model=model, width=640, height=640, conf=0.5, iou=0.5)
        for _ in range(3):
            n = len(det.detect(frame))
        t = time.perf_counter()
        for _ in range(10):
            det.detect(frame)
        dt = (time.perf_counter() - t) / 10
        print("%-44s %7.2f ms per frame (graph path: %s, %d detections)" % (name, 1e3 * dt, det.is_graph(), n), flush=True)
        det.close()
        det = tk.ObjectDetector(model=model, width=640, height=640, conf=0.5, iou=0.5, max_batch=16)
        fr = [frame] * 16
        det.detect_batch(fr)
        t = time.perf_counter()
        for _ in range(5):
            det.detect_batch(fr)
        dt = (time.perf_counter() - t) / 5
        print("%-44s %7.2f ms per call of 16 frames = %.2f ms per frame" % ("", 1e3 * dt, 1e3 * dt / 16), flush=True)
        det.close()
