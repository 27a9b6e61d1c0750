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


def test_decode_policy_restatement_properties():
    """whisper.cpp's decoding policy as the oracle restates it (the reference arms it with 0.2 / 2.4 / -1.0: tk_asr_whisper.c:126-138)"""
    hp = O.WhisperHP(80, 50, 64, 2, 2, 64, 64, 2, 2, 512)
    orc = O.OracleWhisper(hp, seed=6)
    rng = np.random.default_rng(3)
    pcm = np.clip(rng.normal(0, 3000, (2, 8000)), -32768, 32767).astype(np.int16)
    greedy, _, _, lg = orc.transcribe(pcm, 36)
    t0, lp0 = orc.transcribe_policy(pcm, 36, 0.0, 1)
    assert np.array_equal(t0, greedy)                                          # temperature 0 is the greedy decode
    z = lg.astype(np.float64)
    want = -np.log(np.exp(z - z.max(1, keepdims=True)).sum(1))                 # log softmax at the arg max
    assert np.abs(lp0[:, 0] - want).max() < 1e-5
    ta, lpa = orc.transcribe_policy(pcm, 36, 1.0, 7)
    tb, lpb = orc.transcribe_policy(pcm, 36, 1.0, 7)
    tc, _ = orc.transcribe_policy(pcm, 36, 1.0, 8)
    assert np.array_equal(ta, tb) and np.array_equal(lpa, lpb) and not np.array_equal(ta, tc)
    assert (lpa <= 0).all() and lpa.mean() < lp0.mean()                        # drawn tokens are less probable than the arg max
    # the acceptance test on known sequences
    eot = 508
    same = np.full(40, 7, np.int32)
    lp = np.full(40, -0.5, np.float32)
    assert O.whisper_decode_failed(same, lp, eot) == (True, -0.5)              # 32 identical tokens: entropy 0 < 2.4
    distinct = np.arange(40, dtype=np.int32)
    assert O.whisper_decode_failed(distinct, lp, eot) == (False, -0.5)         # entropy ln 32 = 3.47
    assert O.whisper_decode_failed(distinct, lp * 4, eot) == (True, -2.0)      # improbable
    assert O.whisper_decode_failed(same[:32], lp[:32], eot) == (False, -0.5)   # 32 tokens or fewer: no entropy rule
    cut = distinct.copy(); cut[3] = eot
    lp2 = lp.copy(); lp2[:4] = -0.25
    assert O.whisper_decode_failed(cut, lp2 * 8, eot, logprob_thold=-2.5) == (False, -2.0)   # judged up to and including end-of-text only
    half = np.array([i % 4 for i in range(40)], np.int32)                      # four tokens, eight times each: entropy ln 4 = 1.386
    assert O.whisper_decode_failed(half, lp, eot, entropy_thold=1.38)[0] is False and O.whisper_decode_failed(half, lp, eot, entropy_thold=1.39)[0] is True


def test_whisper_logit_filters_on_hand_made_logits():
    """whisper.cpp's whisper_process_logits + the decode loop's token bookkeeping under the reference's parameters
    (/root/reference/src/audio/tk_asr_whisper.c:89-110: suppress_blank off, suppress_non_speech_tokens on, timestamps on, max_initial_ts 1.0),
    restated in orc_whisper_filter_pick, on a toy vocabulary: text 0..19, eot 20, specials 21..29 (always suppressed), timestamps 30..39, the
    first timestamp at most beg + 3.  Every rule is driven by logits built to tempt it."""
    V, EOT, BEG, TID0 = 40, 20, 30, 3
    sup = np.zeros(V, np.uint8)
    sup[21:30] = 1
    sup[7] = 1                                           # a "non-speech" text token
    st = np.zeros(8, np.int32)
    st[7] = 1000                                         # seek_end: a 10 s utterance
    base = np.full(V, -5.0, np.float32)

    def logsoftmax_at(x, allowed, i):
        a = x.astype(np.float64)[allowed]
        return float(x[i] - (np.log(np.exp(a - a.max()).sum()) + a.max()))

    # (a) first token: the best timestamp lies beyond max_initial_ts, the suppressed text token is the largest logit of all, and the timestamps'
    #     summed probability beats every text token -> the best timestamp <= beg + tid0
    x = base.copy(); x[7] = 30.0; x[38] = 10.0; x[32] = 5.0; x[3] = 4.0
    tok, lp = O.whisper_filter_pick(x, sup, st, BEG, EOT, TID0)
    allowed = np.array([i for i in range(V) if not sup[i] and not (i > BEG + TID0)])
    assert tok == 32 and abs(lp - logsoftmax_at(x, allowed, 32)) < 1e-5
    assert st.tolist() == [1, 1, 0, 1, 4, 1, 0, 1000]    # seek_delta 2 * 2, result_len 1, still running
    # (b) the last token was a timestamp and so (fewer than two tokens) was "the one before": no timestamp now, however large
    x = base.copy(); x[35] = 50.0; x[5] = 1.0; x[6] = 0.5
    tok, lp = O.whisper_filter_pick(x, sup, st, BEG, EOT, TID0)
    assert tok == 5 and st.tolist() == [2, 0, 1, 1, 4, 1, 0, 1000]
    # (c) timestamps must not go back: 31 < beg + seek_delta / 2 is masked although it is the largest; 35 wins because the timestamps' sum beats the text
    x = base.copy(); x[31] = 40.0; x[35] = 6.0; x[36] = 5.9; x[4] = 6.2
    tok, lp = O.whisper_filter_pick(x, sup, st, BEG, EOT, TID0)
    assert tok == 35 and st.tolist() == [3, 1, 0, 1, 10, 3, 0, 1000]
    # (d) a timestamp after a text token: only its pair or the end of text may follow -> eot although text logits are larger; the segment is complete
    x = base.copy(); x[2] = 20.0; x[EOT] = 3.0; x[37] = 1.0
    tok, lp = O.whisper_filter_pick(x, sup, st, BEG, EOT, TID0)
    assert tok == EOT and st.tolist() == [4, 0, 1, 1, 10, 3, 1, 1000]
    allowed = np.array([EOT] + list(range(35, 40)))
    assert abs(lp - logsoftmax_at(x, allowed, EOT)) < 1e-5
    # (e) a finished row stands still and emits eot
    before = st.copy()
    assert O.whisper_filter_pick(x, sup, st, BEG, EOT, TID0)[0] == EOT and np.array_equal(st, before)
    # (f) text wins when its best token beats the timestamps' sum: nothing is masked, the arg max is that token
    st = np.zeros(8, np.int32); st[7] = 1000
    x = base.copy(); x[9] = 8.0; x[30] = 5.0; x[31] = 5.0
    assert O.whisper_filter_pick(x, sup, st, BEG, EOT, TID0)[0] == 9 and st.tolist() == [1, 0, 0, 0, 0, 0, 0, 1000]
    # (g) end of text before any timestamp: a failure on a long utterance (whisper.cpp falls back to the next temperature), a one-token result on a
    #     short one (seek + seek_delta + 100 >= seek_end)
    for seek_end, want in ((1000, [1, 0, 0, 0, 0, 0, 2, 1000]), (100, [1, 0, 0, 0, 0, 1, 1, 100])):
        st = np.zeros(8, np.int32); st[7] = seek_end
        x = base.copy(); x[EOT] = 9.0
        assert O.whisper_filter_pick(x, sup, st, BEG, EOT, TID0)[0] == EOT and st.tolist() == want
    # (h) a short utterance ends with its first closing timestamp: seek_delta + 100 >= seek_end
    st = np.array([3, 0, 0, 0, 0, 0, 0, 110], np.int32)
    x = base.copy(); x[36] = 9.0
    assert O.whisper_filter_pick(x, sup, st, BEG, EOT, TID0)[0] == 36 and st.tolist() == [4, 1, 0, 1, 12, 4, 1, 110]
    # (i) at a temperature the draw stays inside the allowed set and is a pure function of (seed, counter)
    st0 = np.array([1, 1, 0, 1, 4, 1, 0, 1000], np.int32)   # after (a): no timestamps allowed
    x = np.linspace(-1, 1, V).astype(np.float32)
    draws = set()
    for counter in range(40):
        s1, s2 = st0.copy(), st0.copy()
        t1, l1 = O.whisper_filter_pick(x, sup, s1, BEG, EOT, TID0, temperature=1.0, seed=5, counter=counter)
        t2, l2 = O.whisper_filter_pick(x, sup, s2, BEG, EOT, TID0, temperature=1.0, seed=5, counter=counter)
        assert (t1, l1) == (t2, l2) and t1 <= EOT and not sup[t1]
        draws.add(t1)
    assert len(draws) > 5
