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


def full_geometry_pcm():
    """tests/golden/make_audio_golden.py: whisper_full_pcm() (pcm_fixture(16000, 1))"""
    rng = np.random.default_rng(2)
    t = np.arange(16000) / 16000.0
    a = 6000 * np.sin(2 * np.pi * 440 * t) + 3000 * np.sin(2 * np.pi * 1250 * t + 0.3) + rng.normal(0, 500, 16000)
    return a[None].astype(np.int16)


def check_against_hf_full_geometry(mel, enc, lg, tok0):
    """sampled log-mel, encoder states and first-step logits of HF transformers at the tiny.en geometry (whisper_full_tiny_en.npz)"""
    g = np.load(os.path.join(GOLD, "whisper_full_tiny_en.npz"))
    assert mel.shape == (1, 3000, 80) and enc.shape == (1, 1500, 384) and lg.shape == (1, 51864)
    assert np.abs(mel[0][g["i_mel"][:, 0], g["i_mel"][:, 1]] - g["hf_mel"]).max() < 2e-4
    assert np.abs(enc[0][g["i_enc"][:, 0], g["i_enc"][:, 1]] - g["hf_enc"]).max() < 2e-4 * max(1.0, float(g["enc_scale"]))
    assert abs(enc.mean() - g["enc_stats"][0]) < 1e-4 and abs(enc.std() - g["enc_stats"][1]) < 1e-4
    assert np.abs(lg[0][g["i_lg"]] - g["hf_lg"]).max() < 2e-4 * max(1.0, float(g["lg_scale"]))
    assert np.abs(lg[0][g["top_ids"]] - g["top_logits"]).max() < 2e-4 * max(1.0, float(g["lg_scale"]))
    assert int(tok0) == int(g["top_ids"][0]) == int(lg[0].argmax())


def test_whisper_oracle_matches_hf_at_the_tiny_en_geometry():
    """the geometry the bench runs (80 mels, 30 s window -> 1500 positions, d 384, 6 heads, 4 + 4 layers, 51864 tokens), not only the toy one:
    the oracle's graph walker (shared with the product) against HF transformers (generator: make_whisper_full)"""
    orc = O.OracleWhisper(O.whisper_tiny_en(), seed=6)
    toks, mel, enc, lg = orc.transcribe(full_geometry_pcm(), 1)
    check_against_hf_full_geometry(mel, enc, lg, toks[0, 0])


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
