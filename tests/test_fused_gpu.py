"""GPU parity for BASELINE configs[3]: the three model streams driven concurrently on their own HIP streams from three host
threads (exactly the structure bench.py times) must give the outputs of the same calls made one at a time."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_llm_group(sess, prompts, n_decode):
    first = sess.prefill(prompts)
    toks, _ = sess.decode(prompts.shape[0], n_decode)
    return first, toks


def test_three_streams_concurrent_equal_standalone(gpu):
    """configs[3]: detector + ASR/VAD + full 32-layer Mistral-7B decode groups, concurrently vs stand-alone: token ids, detection
    lists (class, confidence bits, rect) and ASR token ids identical."""
    G, B, P, N, NF = 2, 32, 8, 8, 8
    model = gpu.LlmModel(gpu.MISTRAL_7B()).fill_synthetic(4)
    hp = model.hparams
    sessions = [gpu.LlmSession(model, B, P + N + 8) for _ in range(G)]
    rng = np.random.default_rng(21)
    prompts = [rng.integers(3, hp.vocab, (B, P)).astype(np.int32) for _ in range(G)]
    for p in prompts:
        p[:, 0] = 1
    det = gpu.ObjectDetector(model="synthetic://yolov8n?seed=5&cls_bias=-0.45", width=640, height=640, conf=0.5, iou=0.5, max_batch=NF)
    asr = gpu.Asr(hp=gpu.WHISPER_TINY_EN(), seed=6, max_batch=NF)
    vad = gpu.Vad()
    frames = [rng.integers(0, 256, (640, 640, 3), dtype=np.uint8) for _ in range(NF)]
    pcm = np.clip(rng.normal(0, 3000, (NF, 16000)), -32768, 32767).astype(np.int16)

    def vision():
        return det.detect_batch(frames)

    def audio():
        ev = []
        for b in range(NF):
            vad.reset()
            ev.append(vad.process_with_events(pcm[b]))
        toks, _, _, _ = asr.transcribe_tokens(pcm, 8, want_aux=False)
        return ev, toks

    # stand-alone, one call after the other
    want_llm = [_run_llm_group(sessions[g], prompts[g], N) for g in range(G)]
    want_det = vision()
    want_ev, want_asr = audio()
    assert sum(len(d) for d in want_det) > 0, "the synthetic detector should fire on random frames (cls_bias=-0.45)"

    # concurrently: G LLM threads + vision + audio, twice (the second round re-uses the captured decode graphs)
    for rnd in range(2):
        got = {}

        def wrap(key, fn, *a):
            def run():
                try:
                    got[key] = fn(*a)
                except Exception as e:  # surfaced below
                    got[key] = e
            return threading.Thread(target=run)

        th = [wrap(("llm", g), _run_llm_group, sessions[g], prompts[g], N) for g in range(G)] + [wrap("det", vision), wrap("aud", audio)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for k, v in got.items():
            if isinstance(v, Exception):
                raise v
        for g in range(G):
            first, toks = got[("llm", g)]
            assert np.array_equal(first, want_llm[g][0]), f"round {rnd} group {g}: first sampled tokens differ under concurrency"
            assert np.array_equal(toks, want_llm[g][1]), f"round {rnd} group {g}: decoded ids differ under concurrency"
        assert len(got["det"]) == NF
        for a, b in zip(got["det"], want_det):
            assert [(c, l, np.float32(s).tobytes(), r) for c, l, s, r in a] == [(c, l, np.float32(s).tobytes(), r) for c, l, s, r in b]
        ev, toks = got["aud"]
        assert ev == want_ev
        assert np.array_equal(toks, want_asr)
    for s in sessions:
        s.close()
    det.close(); asr.close(); vad.close()
    model.close()
