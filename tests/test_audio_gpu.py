"""GPU parity: log-mel, Whisper encoder/decoder tokens and the VAD against the oracle (bit-exact)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def hp_of(gpu, o):
    return gpu.WhisperHP(*[getattr(o, n) for n, _ in o._fields_])


def test_whisper_small_geometry_bit_exact(gpu):
    g = np.load(os.path.join(GOLD, "whisper_tiny.npz"))
    ohp = O.whisper_tiny_test()
    asr = gpu.Asr(hp=hp_of(gpu, ohp), seed=6, max_batch=3)
    orc = O.OracleWhisper(ohp, seed=6)
    rng = np.random.default_rng(12)
    pcm = np.concatenate([g["pcm"], np.clip(rng.normal(0, 3000, (1, 16000)), -32768, 32767).astype(np.int16)])
    toks, mel, enc, lg = asr.transcribe_tokens(pcm, 8)
    wt, wmel, wenc, wlg = orc.transcribe(pcm, 8)
    assert np.array_equal(mel, wmel), np.abs(mel - wmel).max()
    assert np.array_equal(enc, wenc), np.abs(enc - wenc).max()
    assert np.array_equal(lg, wlg), np.abs(lg - wlg).max()
    assert np.array_equal(toks, wt)                                                   # forced-decode ids bit-exact
    assert np.abs(mel[:2] - g["hf_mel"]).max() < 2e-4 and np.array_equal(toks[:2, :6], g["oracle_tokens"])
    # ragged / empty input: 100 samples, and batch of 1
    t1, m1, _, _ = asr.transcribe_tokens(pcm[:1, :100], 2)
    w1, wm1, _, _ = orc.transcribe(pcm[:1, :100], 2)
    assert np.array_equal(t1, w1) and np.array_equal(m1, wm1)


def test_whisper_tiny_en_full_size_encoder(gpu):
    """BASELINE geometry (30 s window, 1500 positions, d 384): mel + encoder + 2 forced steps, one utterance of 1 s"""
    ohp = O.whisper_tiny_en()
    asr = gpu.Asr()                                     # reference surface: tk_asr_whisper_create, synthetic tiny.en
    orc = O.OracleWhisper(ohp, seed=6)
    rng = np.random.default_rng(2)
    pcm = np.clip(rng.normal(0, 3000, (1, 16000)), -32768, 32767).astype(np.int16)
    toks, mel, enc, lg = asr.transcribe_tokens(pcm, 2)
    wt, wmel, wenc, wlg = orc.transcribe(pcm, 2)
    assert np.array_equal(mel, wmel)
    assert np.array_equal(enc, wenc), np.abs(enc - wenc).max()
    assert np.array_equal(toks, wt) and np.array_equal(lg, wlg)


def test_whisper_ggml_checkpoint_end_to_end(gpu, tmp_path):
    """a whisper.cpp ggml .bin (f16 weights) goes in through tk_asr_whisper_create's model_path; geometry, filter bank and vocabulary
    come from the file; mel / encoder / logits / forced ids equal the oracle run on the same (f16-rounded) weights"""
    import ggml_whisper_util as G
    ohp = O.whisper_tiny_test()
    orc = O.OracleWhisper(ohp, seed=9)
    T = orc.tensors()
    rounded, file_t = G.checkpoint_tensors(T, ohp)
    orc.set_tensors(rounded)
    vocab = [(" w%d" % i).encode() for i in range(ohp.n_vocab - 7)]
    path = tmp_path / "ggml-small.bin"
    G.write_ggml(path, ohp, T["frontend.mel_filters"], vocab, file_t)
    asr = gpu.Asr(model=str(path))
    assert [getattr(asr.hp, n) for n, _ in asr.hp._fields_] == [getattr(ohp, n) for n, _ in ohp._fields_]
    rng = np.random.default_rng(3)
    pcm = np.clip(rng.normal(0, 3000, (1, 16000)), -32768, 32767).astype(np.int16)
    toks, mel, enc, lg = asr.transcribe_tokens(pcm, 6)
    wt, wmel, wenc, wlg = orc.transcribe(pcm, 6)
    assert np.array_equal(mel, wmel) and np.array_equal(enc, wenc) and np.array_equal(lg, wlg)
    assert np.array_equal(toks, wt)
    # the reference surface renders ids through the file's vocabulary (specials render as nothing)
    asr.set_decode_steps(6)
    text, n, conf, partial = asr.process_audio(pcm[0], True)
    want = "".join(vocab[t].decode() for t in wt[0] if t < len(vocab))
    eot = ohp.n_vocab - 4
    if eot in list(wt[0]):
        want = "".join(vocab[t].decode() for t in list(wt[0])[: list(wt[0]).index(eot)] if t < len(vocab))
    assert (text or "") == want


def test_asr_reference_surface_buffering(gpu):
    asr = gpu.Asr()
    asr.set_decode_steps(3)
    rng = np.random.default_rng(5)
    chunk = np.clip(rng.normal(0, 3000, 8000), -32768, 32767).astype(np.int16)
    text, n, conf, partial = asr.process_audio(chunk, False)
    assert text is None and n == 0                                   # < 16000 samples and not final: empty result
    text, n, conf, partial = asr.process_audio(chunk, False)
    assert text and partial and abs(conf - 0.9) < 1e-7 and len(text.split()) == 3
    text2, _, _, partial2 = asr.process_audio(chunk[:10], True)      # final: transcribes buffer, then clears it
    assert text2 and not partial2
    t3, n3, _, _ = asr.process_audio(chunk[:10], False)
    assert t3 is None and n3 == 0
    assert gpu.lib().tk_asr_whisper_process_audio(None, None, 0, False, None) == 1001


def test_vad_probabilities_and_events(gpu):
    vad = gpu.Vad()
    rng = np.random.default_rng(3)
    wins = rng.normal(0, 0.1, (40, 480)).astype(np.float32)
    got = vad.probabilities(wins)
    want = O.vad_probabilities(7, wins)
    assert np.array_equal(got, want)
    # state machine through the ABI == fixture traces
    g = json.load(open(os.path.join(GOLD, "vad_state_machine.json")))
    for name, case in g.items():
        vad.reset()
        ev = [[i, e] for i, p in enumerate(case["probabilities"]) for e in [vad.step(p)] if e >= 0]
        assert ev == case["events"], name
    # streaming entry: 1 s of PCM in two ragged chunks == windows 30 ms / hop 10 ms over the joined signal
    vad.reset()
    pcm = np.clip(rng.normal(0, 6000, 16000), -32768, 32767).astype(np.int16)
    ev = vad.process_with_events(pcm[:5000]) + vad.process_with_events(pcm[5000:])
    f = pcm.astype(np.float32) / np.float32(32768.0)
    nwin = (16000 - 480) // 160 + 1
    probs = O.vad_probabilities(7, np.stack([f[k * 160:k * 160 + 480] for k in range(nwin)]))
    wev, ws = O.vad_run(probs)
    assert ev == [e for _, e in wev]
    st = vad.state()
    assert st.is_speech_active == bool(ws.active) and np.float32(st.speech_probability) == probs[-1]
    assert vad.set_threshold(1.5) == 1001 and vad.set_threshold(0.8) == 0
    assert abs(vad.probability(pcm[:480]) - probs[0]) == 0


def test_asr_set_language_selects_the_decoder_prompt(gpu):
    """tk_asr_whisper_set_language (src/audio/tk_asr_whisper.c:377-396): the language reaches the decoder the way whisper.cpp's
    whisper_full uses it — multilingual vocabularies start from <|sot|><|lang|><|transcribe|><|notimestamps|>, English-only ones from
    <|sot|><|notimestamps|> whatever the language; an unknown language fails the next decode, not the setter."""
    ml = gpu.WhisperHP(80, 50, 64, 2, 2, 32, 64, 2, 2, 51865)          # small geometry, multilingual vocabulary size
    asr = gpu.Asr(hp=ml, seed=11)
    assert asr.prompt_tokens() == [50258, 50259, 50359, 50363]          # default: English, transcribe
    assert asr.set_language("de") == 0
    assert asr.prompt_tokens() == [50258, 50261, 50359, 50363]
    rng = np.random.default_rng(4)
    pcm = np.clip(rng.normal(0, 3000, (1, 16000)), -32768, 32767).astype(np.int16)
    toks, mel, enc, lg = asr.transcribe_tokens(pcm, 4)
    orc = O.OracleWhisper(O.WhisperHP(80, 50, 64, 2, 2, 32, 64, 2, 2, 51865), seed=11)
    wt, wmel, wenc, wlg = orc.transcribe(pcm, 4, prompt=[50258, 50261, 50359, 50363])
    assert np.array_equal(toks, wt) and np.array_equal(lg, wlg) and np.array_equal(enc, wenc)
    assert asr.set_language("ja") == 0 and asr.prompt_tokens()[1] == 50258 + 1 + 7
    tj, _, _, lj = asr.transcribe_tokens(pcm, 4)
    wj, _, _, wlj = orc.transcribe(pcm, 4, prompt=[50258, 50266, 50359, 50363])
    assert np.array_equal(tj, wj) and np.array_equal(lj, wlj)
    assert not np.array_equal(lj, lg)                                   # the language token changes the logits
    assert asr.set_language("xx") == 0                                  # stored like the reference stores it...
    assert asr.prompt_tokens() is None
    with pytest.raises(gpu.TkError) as e:                               # ...and the decode fails (whisper_full returns an error)
        asr.transcribe_tokens(pcm, 2)
    assert e.value.code == 4002 and "xx" in e.value.detail
    assert asr.set_language(None) == 1001 and gpu.lib().tk_asr_whisper_set_language(None, b"en") == 1001
    asr.close()
    en = gpu.Asr(hp=gpu.WhisperHP(80, 50, 64, 2, 2, 32, 64, 2, 2, 51864), seed=11)
    assert en.set_language("de") == 0 and en.prompt_tokens() == [50257, 50362]
    en.close()
