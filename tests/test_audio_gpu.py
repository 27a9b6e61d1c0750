"""GPU parity: log-mel, Whisper encoder/decoder tokens and the VAD against the oracle (bit-exact)."""
import ctypes as C
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


def test_whisper_tiny_en_full_size_against_hf(gpu):
    """the HIP ASR engine itself against HF transformers at the tiny.en geometry (whisper_full_tiny_en.npz: sampled log-mel, encoder states,
    first-step logits; 2e-4 of the tensors' scale), without the oracle in between"""
    from test_oracle_audio import check_against_hf_full_geometry, full_geometry_pcm
    asr = gpu.Asr()
    toks, mel, enc, lg = asr.transcribe_tokens(full_geometry_pcm(), 1)
    check_against_hf_full_geometry(mel, enc, lg, toks[0, 0])


def test_asr_batch_composition_invariance_full_size(gpu):
    """size-independent property at the BASELINE geometry: an utterance's mel, encoder states, logits and tokens do not depend on which
    other utterances share its batch (5 utterances: 7500 encoder rows, 30 row blocks of the tiled GEMM; alone: 6 row blocks; the decoder
    runs 1 M-tile instead of 1 — and at 20 utterances 2), for different utterance lengths in one batch"""
    rng = np.random.default_rng(8)
    pcm = np.clip(rng.normal(0, 3000, (5, 16000)), -32768, 32767).astype(np.int16)
    pcm[3, 9000:] = 0                                   # a shorter utterance among full ones
    batch = gpu.Asr(hp=gpu.WHISPER_TINY_EN(), seed=6, max_batch=20)
    toks, mel, enc, lg = batch.transcribe_tokens(pcm, 4)
    for b in (0, 3):
        t1, m1, e1, l1 = batch.transcribe_tokens(pcm[b:b + 1], 4)
        assert np.array_equal(m1[0], mel[b]) and np.array_equal(e1[0].view(np.uint32), enc[b].view(np.uint32))
        assert np.array_equal(l1[0].view(np.uint32), lg[b].view(np.uint32)) and np.array_equal(t1[0], toks[b])
    wide = np.concatenate([pcm] * 4)                    # 20 utterances: the decoder's linears run two M-tiles
    t20, _, _, l20 = batch.transcribe_tokens(wide, 4)
    for r in range(4):
        assert np.array_equal(t20[5 * r:5 * r + 5], toks) and np.array_equal(l20[5 * r:5 * r + 5].view(np.uint32), lg.view(np.uint32))
    batch.close()


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
    asr.set_reference_decode(False)   # the buffering rules, on the forced decode: exactly decode_steps words per result
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


def test_audio_pipeline_vad_asr_flow_matches_the_direct_calls(gpu):
    """tk_audio_pipeline_* (src/audio/tk_audio_pipeline.c:550-803): chunks -> ring -> 32 ms VAD frames -> events -> speech segment -> final
    transcription -> back to AWAITING_WAKE_WORD.  The segment boundaries, events and text equal the same VAD / ASR entry points driven by
    hand on the same frames (process_vad's order: events first, then the frame joins the segment while speech is active)."""
    rng = np.random.default_rng(2)
    loud = np.clip(rng.normal(0, 9000, 16000), -32768, 32767).astype(np.int16)
    sig = np.concatenate([loud, np.zeros(24000, np.int16)])
    ap = gpu.AudioPipeline(threshold=0.8, silence_ms=500.0)                 # no wake-word model: always awake
    assert ap.state() in (1, 2)
    for chunk in np.split(sig, 25):                                          # 100 ms chunks, like the reference's mock microphone
        assert ap.feed(chunk) == 0
        assert ap.drain() == 0
    # by hand
    vad, asr = gpu.Vad(threshold=0.8, min_silence_ms=500.0), gpu.Asr()
    seg, want_events, want_text = [], [], []
    frames = [ch[k:k + 512] for ch in np.split(sig, 25) for k in range(0, len(ch), 512)]   # a drained ring hands out 512, 512, 512, 64 per chunk
    for fr in frames:
        ev = vad.process_with_events(fr)
        want_events += ev
        if 1 in ev and seg:
            text, _, conf, partial = asr.process_audio(np.concatenate(seg), True)
            if text:                                                        # the pipeline reports a transcription only when there is text (:92)
                want_text.append((text, True, conf))
            seg = []
        if vad.state().is_speech_active:
            seg.append(fr)
    assert want_events.count(0) >= 1 and want_events.count(1) >= 1
    assert ap.vad_events == want_events
    assert [(t, f) for t, f, _ in ap.transcriptions] == [(t, f) for t, f, _ in want_text]
    assert all(abs(c - 0.9) < 1e-7 for _, _, c in ap.transcriptions)
    assert ap.state() in (1, 2) and ap.force_end() == 0
    assert ap.feed(np.zeros(20000, np.int16)) == 1004                        # TK_ERROR_BUFFER_TOO_SMALL: larger than the 16384-sample ring
    assert gpu.lib().tk_audio_pipeline_process_chunk(None, None, 0) == 1001
    ap.close()
    # with a wake-word model configured, audio before the detector's hit goes to the detector only
    ap = gpu.AudioPipeline(wake_word="porcupine.pv", threshold=0.8, silence_ms=500.0)
    assert ap.state() == 1
    for chunk in np.split(sig[:16000], 10):
        assert ap.feed(chunk) == 0
    assert ap.drain() == 0 and ap.vad_events == [] and ap.transcriptions == []
    assert ap.wake() == 0 and ap.state() == 2 and ap.wake() == 1002          # TK_ERROR_INVALID_STATE: already listening
    for chunk in np.split(sig, 25):
        assert ap.feed(chunk) == 0
        assert ap.drain() == 0
    assert ap.vad_events == want_events and [(t, f) for t, f, _ in ap.transcriptions] == [(t, f) for t, f, _ in want_text]
    assert ap.state() == 1                                                   # back to waiting for the wake word after the transcription
    ap.close()


def test_audio_pipeline_tts_priority_queue_and_interruption(gpu):
    """the TTS hand-off (tk_audio_pipeline.c:837-1010): requests leave the queue by priority (FIFO inside one), a more urgent request
    interrupts a LOW / NORMAL one that is being spoken (on_tts_interrupt, its remaining audio dropped), never a HIGH one; 16 entries at most."""
    import threading
    import time
    LOW, NORMAL, HIGH, CRITICAL = 3, 2, 1, 0
    ap = gpu.AudioPipeline()
    spoken, gate = [], threading.Event()

    def synth(text, emit):
        spoken.append(text)
        emit(np.full(100, 1, np.int16))
        if text == b"first":
            gate.wait(10)
        emit(np.full(100, 2, np.int16))
        return 0

    ap.set_synthesizer(synth)
    assert ap.say("first", LOW) == 0
    t0 = time.time()
    while not spoken and time.time() - t0 < 10:
        time.sleep(0.005)
    assert spoken == [b"first"] and ap.state() == 4                          # SYNTHESIZING
    for text, pr in (("n1", NORMAL), ("low", LOW), ("h1", HIGH), ("n2", NORMAL), ("crit", CRITICAL), ("h2", HIGH)):
        assert ap.say(text, pr) == 0
    assert ap.interrupts >= 1                                                # NORMAL / HIGH / CRITICAL arrived while a LOW one was speaking
    gate.set()
    assert ap.drain() == 0
    assert spoken == [b"first", b"crit", b"h1", b"h2", b"n1", b"n2", b"low"]
    first_chunks = [a for a, sr in ap.tts_audio[:2]]
    assert first_chunks[0][0] == 1                                           # "first": the chunk before the interruption was delivered ...
    assert sum(1 for a, _ in ap.tts_audio if a[0] == 2) == 6                 # ... its second chunk was dropped; the six others were complete
    # a HIGH request being spoken is not interrupted by a CRITICAL one
    n_int = ap.interrupts
    gate.clear()
    spoken.clear()

    def synth2(text, emit):
        spoken.append(text)
        if text == b"urgent":
            gate.wait(10)
        emit(np.full(10, 7, np.int16))
        return 0

    ap.set_synthesizer(synth2)
    assert ap.say("urgent", HIGH) == 0
    t0 = time.time()
    while not spoken and time.time() - t0 < 10:
        time.sleep(0.005)
    assert ap.say("fire", CRITICAL) == 0 and ap.interrupts == n_int
    for i in range(15):
        assert ap.say("x%d" % i, LOW) == (0 if i < 14 else 1004)             # 16 entries: urgent + fire + 14
    gate.set()
    assert ap.drain() == 0 and spoken[:2] == [b"urgent", b"fire"] and len(spoken) == 16
    assert gpu.lib().tk_audio_pipeline_synthesize_text(ap.h, None, 0) == 1001
    ap.close()


def test_vad_onnx_graph_on_the_gpu_matches_torch(gpu, tmp_path):
    """tk_vad_silero_create on an .onnx model_path (src/sensors/tk_vad_silero.c:110-280): the graph itself — reflect pad, STFT conv,
    magnitude, conv / ReLU encoder, LSTM with recurrent state, 1x1 conv, sigmoid, mean — runs node by node on the GPU
    (csrc/audio/tk_vad_graph.hip).  Probabilities of eight consecutive windows against an independent torch implementation
    (tests/golden/vad_graph.npz), state handling, and the failure mode for an op outside the supported class."""
    import onnx_util
    g = np.load(os.path.join(GOLD, "vad_graph.npz"))
    W = onnx_util.vad_weights(int(g["seed"]))
    path = tmp_path / "silero_class_vad.onnx"
    path.write_bytes(onnx_util.vad_model(W))
    vad = gpu.Vad(model=str(path))
    wins, want = g["windows"], g["torch_probs"]
    got = vad.probabilities(wins)
    assert np.abs(got - want).max() < 2e-5, np.abs(got - want).max()
    # the recurrent state is carried from window to window and from call to call; reset() clears it
    vad.reset()
    again = np.concatenate([vad.probabilities(wins[:3]), vad.probabilities(wins[3:])])
    assert np.array_equal(again, got)
    cont = vad.probabilities(wins[:2])
    assert not np.array_equal(cont, got[:2])
    vad.reset()
    assert np.array_equal(vad.probabilities(wins[:2]), got[:2])
    # the streaming entry drives the same graph: 1 s of the fixture's tone, events from the reference state machine on the graph's output
    pcm = np.clip(np.tile(wins.reshape(-1), 5)[:16000] * 32768.0, -32768, 32767).astype(np.int16)
    vad.reset()
    ev = vad.process_with_events(pcm)
    st = vad.state()
    assert 0.0 < st.speech_probability < 1.0 and all(e in (0, 1) for e in ev)
    # the stand-alone single-window query answers from a cleared state, twice the same
    a, b = vad.probability(pcm[:480]), vad.probability(pcm[:480])
    assert a == b and abs(a - got[0]) < 1e-3     # the same first window, up to the int16 round trip
    vad.close()
    bad = tmp_path / "vad_loop.onnx"
    bad.write_bytes(onnx_util.vad_model(W, extra_op="Loop"))
    with pytest.raises(gpu.TkError) as e:
        gpu.Vad(model=str(bad))
    assert e.value.code == 4000 and "Loop" in e.value.detail
    # the per-sample-rate switch of Silero-class exports: If on Equal(sr, 16000) with the head inside the branches.  At 16 kHz the then
    # branch runs — the torch fixture's arithmetic; at 8 kHz the else branch — the same numbers as that arithmetic written as a plain graph
    sw = tmp_path / "vad_if.onnx"
    sw.write_bytes(onnx_util.vad_model(W, with_if=True))
    v16 = gpu.Vad(model=str(sw))
    assert np.array_equal(v16.probabilities(wins), got)
    v16.close()
    sw8 = tmp_path / "vad_if_8k.onnx"
    sw8.write_bytes(onnx_util.vad_model(W, window=240, with_if=True))
    plain8 = tmp_path / "vad_neg_8k.onnx"
    plain8.write_bytes(onnx_util.vad_model(W, window=240, neg_head=True))
    w8 = wins.reshape(-1, 240)
    a8, b8 = gpu.Vad(model=str(sw8), sample_rate=8000), gpu.Vad(model=str(plain8), sample_rate=8000)
    pa, pb = a8.probabilities(w8), b8.probabilities(w8)
    assert np.array_equal(pa, pb) and pa.std() > 1e-4
    a8.close(); b8.close()


def _policy_geometry():
    # the small test geometry with a 64-position text context: room for more than 32 decoded tokens (the entropy rule's window)
    return O.WhisperHP(80, 50, 64, 2, 2, 64, 64, 2, 2, 512)


def test_asr_decode_policy_tokens_and_logprobs_bit_exact(gpu):
    """whisper.cpp's per-step bookkeeping (the reference arms its policy: tk_asr_whisper.c:126-138): tokens picked by temperature and their
    log-probabilities, device against oracle, bit for bit; temperature 0 is the greedy decode"""
    ohp = _policy_geometry()
    asr = gpu.Asr(hp=hp_of(gpu, ohp), seed=6, max_batch=3)
    orc = O.OracleWhisper(ohp, seed=6)
    rng = np.random.default_rng(21)
    pcm = np.clip(rng.normal(0, 3000, (3, 16000)), -32768, 32767).astype(np.int16)
    greedy, _, _, lg = asr.transcribe_tokens(pcm, 40)
    for temp, seed in ((0.0, 0), (0.4, 11), (1.0, 12), (1.0, 13)):
        toks, lp = asr.transcribe_policy(pcm, 40, temp, seed)
        wt, wlp = orc.transcribe_policy(pcm, 40, temp, seed)
        assert np.array_equal(toks, wt), (temp, seed)
        assert np.array_equal(lp.view(np.uint32), wlp.view(np.uint32)), (temp, np.abs(lp - wlp).max())
        assert (lp <= 0).all() and np.isfinite(lp).all()
        if temp == 0.0:
            assert np.array_equal(toks, greedy)
            z = lg.astype(np.float64)
            ref = z.max(1) - np.log(np.exp(z - z.max(1, keepdims=True)).sum(1)) - z.max(1)      # log softmax at the arg max
            assert np.abs(lp[:, 0] - ref).max() < 1e-5
    a, _ = asr.transcribe_policy(pcm, 40, 1.0, 12)
    b, _ = asr.transcribe_policy(pcm, 40, 1.0, 13)
    assert not np.array_equal(a, b) and not np.array_equal(a, greedy)                          # the draws are real, and keyed by the seed
    one, lp1 = asr.transcribe_policy(pcm[:1], 40, 1.0, 12)                                     # batch 1: counters are position x batch + row
    w1, wlp1 = orc.transcribe_policy(pcm[:1], 40, 1.0, 12)
    assert np.array_equal(one, w1) and np.array_equal(lp1.view(np.uint32), wlp1.view(np.uint32))
    assert gpu.lib().tk_mi355x_asr_transcribe_policy(asr.h, 1, pcm.ctypes.data_as(C.c_void_p), 16000, 4, C.c_float(-0.5), C.c_uint64(0),
                                                     one.ctypes.data_as(C.c_void_p), None) != 0   # negative temperature refused


def _fallback(orc, pcm, n_steps, eot, inc, ent, lpt, seed, is_final=True):
    """whisper.cpp's temperature ladder, restated on the oracle's decodes"""
    t, attempts = np.float32(0.0), 0
    while True:
        toks, lp = orc.transcribe_policy(pcm[None], n_steps, float(t), seed + attempts)
        attempts += 1
        failed, avg = O.whisper_decode_failed(toks[0], lp[0], eot, ent, lpt)
        nxt = np.float32(t + np.float32(inc))
        if not failed or not is_final or not inc > 0 or not nxt < np.float32(np.float32(1.0) + np.float32(1e-6)):
            return toks[0], float(t), avg, attempts
        t = nxt


def test_asr_decode_policy_fallback_on_the_reference_surface(gpu):
    """tk_asr_whisper_process_audio with the policy on walks the temperature ladder exactly as the restated loop over oracle decodes does;
    with the policy off (the default) it is the plain greedy decode"""
    ohp = _policy_geometry()
    asr = gpu.Asr(hp=hp_of(gpu, ohp), seed=6, max_batch=1)
    orc = O.OracleWhisper(ohp, seed=6)
    asr.set_decode_steps(40)
    eot = ohp.n_vocab - 4
    rng = np.random.default_rng(22)
    pcm = np.clip(rng.normal(0, 3000, 16000), -32768, 32767).astype(np.int16)

    def text_of(toks):
        out = ""
        for t in toks:
            if t == eot:
                break
            out += " w%d" % t
        return out

    greedy, _, _, _ = asr.transcribe_tokens(pcm[None], 40, want_aux=False)
    text, _, _, _ = asr.process_audio(pcm, True)
    assert text == text_of(greedy[0]) and asr.last_decode() == (0.0, 0.0, 0)                  # default: off
    # averages of the ladder's rungs, to place thresholds between them
    avgs = []
    for a in range(6):
        toks, lp = orc.transcribe_policy(pcm[None], 40, float(np.float32(0.2) * a), 5 + a)
        avgs.append(O.whisper_decode_failed(toks[0], lp[0], eot, 0.0, -1e9)[1])
    cases = [(0.2, 2.4, -1.0), (0.2, 0.0, -1e9), (0.2, 1e9, -1e9), (0.2, 0.0, float(np.median(avgs))), (0.5, 0.0, 0.0), (-1.0, 1e9, 0.0)]
    seen = set()
    for inc, ent, lpt in cases:
        asr.set_decode_policy(True, inc, ent, lpt, seed=5)
        text, _, conf, partial = asr.process_audio(pcm, True)
        wt, wtemp, wavg, watt = _fallback(orc, pcm, 40, eot, inc, ent, lpt, 5)
        temp, avg, att = asr.last_decode()
        assert text == text_of(wt), (inc, ent, lpt)
        assert (np.float32(temp), att) == (np.float32(wtemp), watt) and np.float32(avg) == np.float32(wavg), (inc, ent, lpt, temp, avg, att, wtemp, wavg, watt)
        assert not partial and abs(conf - 0.9) < 1e-7
        seen.add(att)
    assert 1 in seen and 6 in seen                                                            # accepted at once, and the whole ladder
    # a partial result decodes once whatever the thresholds
    asr.set_decode_policy(True, 0.2, 1e9, 0.0, seed=5)
    text, _, _, partial = asr.process_audio(pcm, False)
    wt, _, _, watt = _fallback(orc, pcm, 40, eot, 0.2, 1e9, 0.0, 5, is_final=False)
    assert partial and text == text_of(wt) and asr.last_decode()[2] == 1 == watt
    asr.reset()
    asr.set_decode_policy(False)
    text, _, _, _ = asr.process_audio(pcm, True)
    assert text == text_of(greedy[0])


def test_asr_decode_policy_from_the_environment(gpu, monkeypatch):
    """TK_MI355X_ASR_POLICY=1: tk_asr_whisper_create arms the policy with the reference's numbers (0.2 / 2.4 / -1.0, tk_asr_whisper.c:126-138)"""
    rng = np.random.default_rng(23)
    pcm = np.clip(rng.normal(0, 3000, 16000), -32768, 32767).astype(np.int16)
    plain = gpu.Asr()
    plain.set_decode_steps(4)
    plain.process_audio(pcm, True)
    assert plain.last_decode()[2] == 0
    monkeypatch.setenv("TK_MI355X_ASR_POLICY", "1")
    armed = gpu.Asr()
    armed.set_decode_steps(4)
    armed.process_audio(pcm, True)
    temp, avg, attempts = armed.last_decode()
    assert attempts >= 1 and avg < 0.0
    # synthetic weights give near-uniform logits: mean log-probability far below -1, so the whole ladder runs and ends at temperature 1
    assert attempts == 6 and abs(temp - 1.0) < 1e-5
    armed.process_audio(pcm, False)                                   # partial: one decode
    assert armed.last_decode()[2] == 1
    plain.close(); armed.close()


def test_asr_contexts_on_one_file_share_an_engine_and_give_the_solo_results(gpu):
    """round 6: tk_asr_whisper_context_t handles opened on the same checkpoint share the weights and ONE batched engine; their
    one-utterance calls (src/audio/tk_asr_whisper.c:282-344) are coalesced by a scheduler, shorter utterances padded with the zeros the 30 s
    window holds behind them anyway.  Five contexts fed utterances of DIFFERENT lengths at once return the text each returns alone."""
    import threading
    rng = np.random.default_rng(78)
    K = 5
    utts = [np.clip(rng.normal(0, 3000, 16000 + 1777 * i), -32768, 32767).astype(np.int16) for i in range(K)]
    solo = []
    for u in utts:
        a = gpu.Asr()
        a.set_reference_decode(False)   # six words per utterance whatever the (random-weight) decoder says; the default decode: test_asr_reference_parameter_decode
        a.set_decode_steps(6)
        assert a.share_stats()[0] == 1
        solo.append(a.process_audio(u, True)[0])
        a.close()
    assert len(set(solo)) > 1
    ctx = [gpu.Asr() for _ in range(K)]
    for a in ctx:
        a.set_reference_decode(False)
        a.set_decode_steps(6)
    assert ctx[0].share_stats()[0] == K
    got = [None] * K
    bar = threading.Barrier(K)

    def run(i):
        bar.wait()
        for rep in range(2):
            got[i] = ctx[i].process_audio(utts[i], True)[0]

    th = [threading.Thread(target=run, args=(i,)) for i in range(K)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert got == solo
    handles, batches, n, widest = ctx[0].share_stats()
    assert n == 2 * K and batches < n and widest >= 2, (handles, batches, n, widest)
    # the policy decode (temperature ladder) goes through the same engine one utterance at a time and equals a private context's
    ctx[0].set_decode_policy(True, seed=3)
    priv = gpu.Asr()
    priv_alone = None
    for a in ctx[1:]:
        a.close()
    t_shared = ctx[0].process_audio(utts[0], True)[0]
    ctx[0].close()
    priv.set_reference_decode(False)
    priv.set_decode_steps(6)
    priv.set_decode_policy(True, seed=3)
    assert priv.process_audio(utts[0], True)[0] == t_shared
    priv.close()


NON_SPEECH = ['"', "#", "(", ")", "*", "+", "/", ":", ";", "<", "=", ">", "@", "[", "\\", "]", "^", "_", "`", "{", "|", "}", "~", "「", "」", "『", "』", "<<", ">>", "<<<",
              ">>>", "--", "---", "-(", "-[", "('", '("', "((", "))", "(((", ")))", "[[", "]]", "{{", "}}", "♪♪", "♪♪♪", "♩", "♪", "♫", "♬", "♭", "♮", "♯"]


def test_asr_reference_parameter_decode(gpu, tmp_path):
    """tk_asr_whisper_process_audio decodes the way whisper_full does under the parameters the reference's wrapper sets
    (/root/reference/src/audio/tk_asr_whisper.c:89-110: suppress_blank off, suppress_non_speech_tokens on, timestamps on; :142-181: segment texts
    concatenated).  A ggml checkpoint with whisper's real vocabulary layout (51864 ids: text, eot 50256, sot, language / task / special tokens,
    timestamps from 50363) and a small decoder: the product's suppression table equals the one built here from the vocabulary strings, tokens,
    log-probabilities, result lengths and statuses of utterances of several lengths equal the oracle's (orc_whisper_transcribe_ref) — alone,
    coalesced by the shared engine, and at a temperature — and process_audio's text is the text tokens of the accepted run."""
    import threading
    import ggml_whisper_util as G
    ohp = O.WhisperHP(80, 1500, 64, 2, 1, 448, 64, 2, 1, 51864)
    orc = O.OracleWhisper(ohp, seed=12)
    T = orc.tensors()
    rounded, file_t = G.checkpoint_tensors(T, ohp)
    orc.set_tensors(rounded)
    EOT, SOT, BEG = 50256, 50257, 50363
    vocab = [(" w%d" % i).encode() for i in range(EOT)]
    planted = {}
    for k, tok in enumerate(NON_SPEECH + [" -", " '"]):                     # plant the strings whisper.cpp looks up, bare and " "-prefixed
        for j, form in enumerate(([tok, " " + tok] if tok not in (" -", " '") else [tok])):
            i = 100 + 7 * (2 * k + j)
            vocab[i] = form.encode()
            planted[i] = form
    vocab[90] = b" ("                                                        # a second id of the same string: whisper.cpp's map keeps one, so does the table
    path = tmp_path / "ggml-ref.bin"
    G.write_ggml(path, ohp, T["frontend.mel_filters"], vocab, file_t)
    asr = gpu.Asr(model=str(path))
    tab, beg, eot = asr.suppress_table()
    want_tab = np.zeros(ohp.n_vocab, np.uint8)
    want_tab[SOT:BEG] = 1                                                    # sot, 99 language slots, translate, transcribe, solm, prev, nosp, notimestamps
    for i in planted:
        want_tab[i] = 1
    want_tab[100 + 7 * (2 * NON_SPEECH.index("(") + 1)] = 0                  # " (" is found at its FIRST id (90), the planted copy stays free
    want_tab[90] = 1
    assert (beg, eot) == (BEG, EOT) and np.array_equal(tab, want_tab), np.nonzero(tab != want_tab)
    rng = np.random.default_rng(41)
    lens = [16000, 16000 + 3000, 5 * 16000, 25 * 16000, 12 * 16000 + 123]
    utts = [np.clip(rng.normal(0, 3000, n), -32768, 32767).astype(np.int16) for n in lens]
    prompt = np.array([SOT], np.int32)                                       # English-only vocabulary: <|startoftranscript|> alone, timestamps on
    STEPS = 20
    want = []
    for u in utts:
        want.append(orc.transcribe_ref(u[None], [len(u)], STEPS, want_tab, BEG, EOT, prompt))
    statuses = [int(w[3][0]) for w in want]
    for u, w in zip(utts, want):                                             # one at a time
        toks, lp, rl, st = asr.transcribe_ref(u, STEPS)
        assert np.array_equal(toks, w[0][0]) and np.array_equal(lp, w[1][0]) and (rl, st) == (int(w[2][0]), int(w[3][0]))
    assert len(set(statuses)) > 1 or max(int(w[2][0]) for w in want) > 2, statuses   # the cases are not all the same one-token story
    # every rule leaves its trace: no suppressed token is ever sampled, timestamps never decrease
    for w in want:
        t = w[0][0][: int(w[2][0])]
        assert not want_tab[t].any()
        ts = t[t >= BEG]
        assert np.all(np.diff(ts) >= 0)
    # coalesced: five contexts of the same file at once, different lengths in one batch (each row its own seek_end)
    ctx = [gpu.Asr(model=str(path)) for _ in utts]
    got = [None] * len(utts)
    bar = threading.Barrier(len(utts))

    def run(i):
        bar.wait()
        got[i] = ctx[i].transcribe_ref(utts[i], STEPS)

    th = [threading.Thread(target=run, args=(i,)) for i in range(len(utts))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for g, w in zip(got, want):
        assert np.array_equal(g[0], w[0][0]) and np.array_equal(g[1], w[1][0]) and (g[2], g[3]) == (int(w[2][0]), int(w[3][0]))
    assert ctx[0].share_stats()[3] >= 2
    # a long cap (whisper.cpp's own is n_text_ctx / 2 - 4 = 220): every 16 steps the engine reads the rows' status words and stops enqueueing once all
    # rows have ended; tokens behind a row's end are eot and its log-probabilities 0 on both sides
    for i in (0, 2):
        w48 = orc.transcribe_ref(utts[i][None], [len(utts[i])], 48, want_tab, BEG, EOT, prompt)
        g48 = asr.transcribe_ref(utts[i], 48)
        assert np.array_equal(g48[0], w48[0][0]) and np.array_equal(g48[1], w48[1][0]) and (g48[2], g48[3]) == (int(w48[2][0]), int(w48[3][0]))
    # at a temperature: the canonical draw over the allowed set, same seed and counter on both sides
    wt = orc.transcribe_ref(utts[2][None], [len(utts[2])], STEPS, want_tab, BEG, EOT, prompt, temperature=0.6, seed=9)
    gt = asr.transcribe_ref(utts[2], STEPS, temperature=0.6, seed=9)
    assert np.array_equal(gt[0], wt[0][0]) and np.array_equal(gt[1], wt[1][0]) and (gt[2], gt[3]) == (int(wt[2][0]), int(wt[3][0]))
    # the reference surface: the text tokens of the accepted run, specials and timestamps rendered as nothing; less than a second of audio gives no text
    asr.set_decode_steps(STEPS)
    for u, w in zip(utts, want):
        text = asr.process_audio(u, True)[0] or ""
        run_ = w[0][0][: int(w[2][0])]
        assert text == "".join(vocab[t].decode() for t in run_ if t < EOT)
    assert (asr.process_audio(utts[0][:15000], True)[0] or "") == ""
    for a in ctx + [asr]:
        a.close()


def test_asr_fast_contraction_gate(gpu):
    """the opt-in fast contraction (tk_mi355x_asr_set_fast_contraction: log-mel and the encoder's long passes on the f16 matrix pipe with split
    operands) against the exact path and the HF fixture at the tiny.en geometry — VERDICT r05 item 8's gate: log-mel within 1e-4, encoder states and
    first-step logits within 1e-5 of their scale of the exact path and inside the HF fixture's tolerance, forced-decode ids equal; switched
    off again the context returns the exact path's bits."""
    from test_oracle_audio import check_against_hf_full_geometry, full_geometry_pcm
    asr = gpu.Asr()
    pcm = full_geometry_pcm()
    te, me, ee, le = asr.transcribe_tokens(pcm, 8)
    asr.set_fast_contraction(True)
    tf, mf, ef, lf = asr.transcribe_tokens(pcm, 8)
    assert not np.array_equal(ef, ee)                                          # the other kernels did run
    assert np.abs(mf - me).max() < 1e-4
    assert np.abs(ef - ee).max() < 1e-5 * np.abs(ee).max() and np.abs(lf - le).max() < 1e-5 * np.abs(le).max()
    assert np.array_equal(tf, te)
    check_against_hf_full_geometry(mf, ef, lf, tf[0, 0])
    rng = np.random.default_rng(12)                                            # a batch: three clips, one shorter
    clips = np.clip(rng.normal(0, 3000, (3, 16000)), -32768, 32767).astype(np.int16)
    clips[1, 7000:] = 0
    batch = gpu.Asr(hp=gpu.WHISPER_TINY_EN(), seed=6, max_batch=4)
    t0 = batch.transcribe_tokens(clips, 8, want_aux=False)[0]
    batch.set_fast_contraction(True)
    assert np.array_equal(batch.transcribe_tokens(clips, 8, want_aux=False)[0], t0)
    batch.close()
    asr.set_fast_contraction(False)
    t2, m2, e2, l2 = asr.transcribe_tokens(pcm, 8)
    assert np.array_equal(m2, me) and np.array_equal(e2, ee) and np.array_equal(l2, le) and np.array_equal(t2, te)
    asr.close()


def test_asr_contexts_with_and_without_fast_contraction_share_an_engine(gpu):
    """contexts of one checkpoint may differ in the opt-in fast contraction: their utterances ride separate jobs of the shared engine, and every context
    returns what it returns alone — four contexts, two of them fast, fed from four threads at once"""
    import threading
    rng = np.random.default_rng(79)
    K = 4
    utts = [np.clip(rng.normal(0, 3000, 16000 + 1200 * i), -32768, 32767).astype(np.int16) for i in range(K)]

    def make(i):
        a = gpu.Asr()
        a.set_reference_decode(False)
        a.set_decode_steps(6)
        a.set_fast_contraction(i % 2 == 1)
        return a
    solo = []
    for i in range(K):
        a = make(i)
        solo.append(a.process_audio(utts[i], True)[0])
        a.close()
    ctx = [make(i) for i in range(K)]
    got = [None] * K
    bar = threading.Barrier(K)

    def run(i):
        bar.wait()
        for _ in range(2):
            got[i] = ctx[i].process_audio(utts[i], True)[0]

    th = [threading.Thread(target=run, args=(i,)) for i in range(K)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert got == solo
    handles, batches, n, widest = ctx[0].share_stats()
    assert handles == K and n == 2 * K and batches >= 2                          # at least one job per setting
    for a in ctx:
        a.close()
