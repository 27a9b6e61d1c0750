#!/usr/bin/env python3
"""Stand-alone timing of the perception streams (developer tool, needs an MI355X): detector, VAD, ASR per batch of 32."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trackiellm_amd as tk  # noqa: E402

PB = 32
det = tk.ObjectDetector(model="synthetic://yolov8n?seed=5&cls_bias=-0.45", width=640, height=640, conf=0.5, iou=0.5, device=0, max_batch=PB)
if os.environ.get("TK_PERC_FAST") == "1":  # the opt-in split-f16 contraction (tools/time_fast_perception.py compares it with the exact path)
    det.set_fast_contraction(True)
asr = tk.Asr(hp=tk.WHISPER_TINY_EN(), seed=6, device=0, max_batch=PB)
if os.environ.get("TK_PERC_FAST") == "1":
    asr.set_fast_contraction(True)
vad = tk.Vad()
rng = np.random.default_rng(1)
frames = [rng.integers(0, 256, (640, 640, 3), dtype=np.uint8) for _ in range(PB)]
pcm = np.clip(rng.normal(0, 3000, (PB, 16000)), -32768, 32767).astype(np.int16)


def timed(name, fn, n=3):
    fn()
    t = time.time()
    for _ in range(n):
        fn()
    print(f"{name:28s} {1000 * (time.time() - t) / n:9.2f} ms per {PB}", flush=True)


def vad_pass():
    for b in range(PB):
        vad.reset()
        vad.process_with_events(pcm[b])


ONLY = os.environ.get("TK_PERC_ONLY", "")  # "det", "vad", "asr16", "asr1": one stream only (kernel traces)
if ONLY in ("", "det"):
    timed("detector (32 frames)", lambda: det.detect_batch(frames))
if ONLY in ("", "vad"):
    timed("vad (32 x 1 s)", vad_pass)
if ONLY in ("", "asr16"):
    timed("asr 16 steps (32 x 1 s)", lambda: asr.transcribe_tokens(pcm, 16, want_aux=False))
if ONLY in ("", "asr1"):
    timed("asr 1 step (32 x 1 s)", lambda: asr.transcribe_tokens(pcm, 1, want_aux=False))
if ONLY:
    sys.exit(0)

# depth: a graph of the convolutional MiDaS class (tests/onnx_util.depth_spec at 4x the fixture's channel counts), one 640x480 frame -> 256x256 map
import tempfile  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import onnx_util as OX  # noqa: E402

with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "depth_w4.onnx")
    open(path, "wb").write(OX.depth_model(OX.depth_weights(11, 4), channels=4))
    est = tk.DepthEstimator(path, 256, 256)
    frame = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    est.estimate(frame)
    t = time.time()
    for _ in range(10):
        est.estimate(frame)
    print(f"{'depth (1 frame, 256x256)':28s} {1000 * (time.time() - t) / 10:9.2f} ms per frame ({len(OX.depth_spec(4))} graph nodes)", flush=True)
    est.close()
