#!/usr/bin/env python3
"""The opt-in fast contraction of the perception streams against the exact path (developer tool, needs an MI355X): time per 32 and the
tolerance gate of VERDICT r05 item 8 on the bench's frames — detection indices identical, boxes within 1e-3 of the frame scale, head maps
within a stated fraction of their scale.    python tools/time_fast_perception.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trackiellm_amd as tk  # noqa: E402

PB = 32
rng = np.random.default_rng(1)
frames = [rng.integers(0, 256, (640, 640, 3), dtype=np.uint8) for _ in range(PB)]


def timed(name, fn, n=5):
    fn()
    t = time.time()
    for _ in range(n):
        fn()
    ms = 1000 * (time.time() - t) / n
    print(f"{name:44s} {ms:9.2f} ms per {PB}", flush=True)
    return ms


det = tk.ObjectDetector(model="synthetic://yolov8n?seed=5&cls_bias=-0.45", width=640, height=640, conf=0.5, iou=0.5, device=0, max_batch=PB)
exact = det.detect_batch(frames)
t_exact = timed("detector, exact fp32 chain", lambda: det.detect_batch(frames))
x = rng.random((2, 640, 640, 3), dtype=np.float32)
raw_exact = det.forward_raw(x)
det.set_fast_contraction(True)
fast = det.detect_batch(frames)
t_fast = timed("detector, split-f16 contraction", lambda: det.detect_batch(frames))
raw_fast = det.forward_raw(x)
scale = float(np.abs(raw_exact).max())
print(f"head maps: max |fast - exact| = {np.abs(raw_fast - raw_exact).max():.3e} on a scale of {scale:.3f} ({np.abs(raw_fast - raw_exact).max() / scale:.2e} of scale)")
n_det = sum(len(e) for e in exact)
same_idx = all(len(e) == len(f) and all(a[0] == b[0] for a, b in zip(e, f)) for e, f in zip(exact, fast))
dbox = 0.0
dscore = 0.0
if same_idx:
    for e, f in zip(exact, fast):
        for a, b in zip(e, f):
            dbox = max(dbox, max(abs(p - q) for p, q in zip(a[3], b[3])))  # integer rectangles (the reference's tk_rect_t): 0 or a flip across a pixel border
            dscore = max(dscore, abs(a[2] - b[2]))
if not same_idx:  # say what differs: a detection gained / lost at the threshold, or two neighbours in score order swapped
    for i, (e, f) in enumerate(zip(exact, fast)):
        if len(e) != len(f) or any(a[0] != b[0] for a, b in zip(e, f)):
            se, sf = {(a[0], a[3]) for a in e}, {(b[0], b[3]) for b in f}
            print(f"  frame {i}: {len(e)} / {len(f)} detections, only in exact {sorted(se - sf)[:3]}, only in fast {sorted(sf - se)[:3]}"
                  f"{' (same set, order differs)' if se == sf else ''}")
print(f"detections over {PB} frames: {n_det}; same count / order / classes: {same_idx}; max box difference {dbox:.4f} px (gate: <= {1e-3 * 640:.2f}), max score difference {dscore:.2e}")
print(f"speed-up {t_exact / t_fast:.2f} x")

# ---- ASR: Whisper tiny.en geometry, 32 clips of 1 s, forced decode of 16 steps (the bench's unit) ----
asr = tk.Asr(hp=tk.WHISPER_TINY_EN(), seed=6, device=0, max_batch=PB)
pcm = np.clip(np.random.default_rng(2).normal(0, 3000, (PB, 16000)), -32768, 32767).astype(np.int16)
tok_e, mel_e, enc_e, lg_e = asr.transcribe_tokens(pcm[:4], 16, want_aux=True)
tok32_e = asr.transcribe_tokens(pcm, 16, want_aux=False)[0]
t1_e = timed("asr 1 step, exact fp32 chains", lambda: asr.transcribe_tokens(pcm, 1, want_aux=False), 3)
t16_e = timed("asr 16 steps, exact fp32 chains", lambda: asr.transcribe_tokens(pcm, 16, want_aux=False), 3)
asr.set_fast_contraction(True)
tok_f, mel_f, enc_f, lg_f = asr.transcribe_tokens(pcm[:4], 16, want_aux=True)
tok32_f = asr.transcribe_tokens(pcm, 16, want_aux=False)[0]
t1_f = timed("asr 1 step, split-f16 long passes", lambda: asr.transcribe_tokens(pcm, 1, want_aux=False), 3)
t16_f = timed("asr 16 steps, split-f16 long passes", lambda: asr.transcribe_tokens(pcm, 16, want_aux=False), 3)


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


print(f"log-mel max |fast - exact| {np.abs(mel_f - mel_e).max():.2e} (gate <= 1e-4); encoder output {rel(enc_f, enc_e):.2e} of scale; first-step logits {rel(lg_f, lg_e):.2e} of scale")
print(f"forced-decode ids equal: 4 clips {bool((tok_f == tok_e).all())}, 32 clips {bool((tok32_f == tok32_e).all())} ({int((tok32_f != tok32_e).sum())} of {tok32_e.size} differ)")
print(f"speed-up 1 step {t1_e / t1_f:.2f} x, 16 steps {t16_e / t16_f:.2f} x")
