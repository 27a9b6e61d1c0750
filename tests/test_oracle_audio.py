"""CPU: ASR / VAD oracle against its pins (HF transformers fixture, hand-checked VAD traces)."""
import json
import os

import numpy as np

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_whisper_oracle_matches_hf_fixture():
    g = np.load(os.path.join(GOLD, "whisper_tiny.npz"))
    orc = O.OracleWhisper(O.whisper_tiny_test(), seed=6)
    toks, mel, enc, lg = orc.transcribe(g["pcm"], g["oracle_tokens"].shape[1])
    assert np.array_equal(toks, g["oracle_tokens"]) and np.array_equal(mel, g["oracle_mel"]) and np.array_equal(lg, g["oracle_logits"])
    assert np.abs(mel - g["hf_mel"]).max() < 2e-4                                      # vs HF WhisperFeatureExtractor
    assert np.abs(enc - g["hf_enc"]).max() < 2e-4 * max(1.0, np.abs(g["hf_enc"]).max())  # vs HF WhisperModel encoder
    assert np.abs(lg - g["hf_logits"]).max() < 2e-4 * max(1.0, np.abs(g["hf_logits"]).max())
    assert np.array_equal(g["hf_logits"].argmax(1), toks[:, 0])


def test_mel_edge_cases():
    orc = O.OracleWhisper(O.whisper_tiny_test(), seed=6)
    z = np.zeros((1, 16000), np.int16)                       # the reference test's silence (tests/tk_cortex_test.cpp:90)
    _, mel, _, _ = orc.transcribe(z, 0)
    assert np.all(mel == mel.flat[0]) and abs(mel.flat[0] - (-10 + 4) / 4) < 1e-6      # log10(1e-10) clamps to -10
    short = np.full((1, 100), 1000, np.int16)                # ragged: shorter than the window, zero padded
    _, mel2, _, _ = orc.transcribe(short, 0)
    assert mel2[0, 0].max() > mel2[0, -1].max()


def test_vad_state_machine_fixture():
    g = json.load(open(os.path.join(GOLD, "vad_state_machine.json")))
    for name, case in g.items():
        ev, _ = O.vad_run(case["probabilities"])
        assert [list(e) for e in ev] == case["events"], name
    ev, s = O.vad_run([0.9] * 30, threshold=0.8, min_silence_ms=500.0)                  # cortex settings (tk_cortex_main.c:881-882)
    assert ev == [(8, 0)] and s.active == 1
