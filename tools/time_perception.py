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
asr = tk.Asr(hp=tk.WHISPER_TINY_EN(), seed=6, device=0, max_batch=PB)
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


timed("detector (32 frames)", lambda: det.detect_batch(frames))
timed("vad (32 x 1 s)", vad_pass)
timed("asr 16 steps (32 x 1 s)", lambda: asr.transcribe_tokens(pcm, 16, want_aux=False))
timed("asr 1 step (32 x 1 s)", lambda: asr.transcribe_tokens(pcm, 1, want_aux=False))
