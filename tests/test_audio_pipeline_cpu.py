"""CPU: the audio pipeline's host logic (ring, worker thread, wake-word / listening / transcribing states, TTS priority queue and
interruption — csrc/abi/tk_abi_audio_pipeline.cpp restating src/audio/tk_audio_pipeline.c:387-1010) against stub VAD / ASR engines
(tests/stubs/audio_engine_stub.cpp: amplitude threshold, "seg<N>" transcriptions).  The same flows run against the GPU engines in
tests/test_audio_gpu.py."""
import ctypes as C
import os
import subprocess
import threading
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOW, NORMAL, HIGH, CRITICAL = 3, 2, 1, 0


@pytest.fixture(scope="module")
def stub_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("stub") / "libtk_audio_pipeline_stub.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "stubs", "audio_engine_stub.cpp"),
                           os.path.join(ROOT, "trackiellm_amd", "csrc", "abi", "tk_abi_audio_pipeline.cpp"), "-o", out, "-lpthread", "-Wl,-Bsymbolic"])  # -Bsymbolic: bind to the stubs even when the product library is loaded RTLD_GLOBAL
    return C.CDLL(out)


def make(stub_lib, **kw):
    from trackiellm_amd.audio import AudioPipeline
    return AudioPipeline(library=stub_lib, **kw)


SIG = np.concatenate([np.full(16000, 5000, np.int16), np.zeros(24000, np.int16)])


def test_always_awake_flow(stub_lib):
    ap = make(stub_lib)
    for chunk in np.split(SIG, 25):
        assert ap.feed(chunk) == 0 and ap.drain() == 0
    # the stub VAD starts with the first loud frame and ends with the first silent one: the segment is the loud second
    assert ap.vad_events == [0, 1] and ap.transcriptions == [("seg16000", True, pytest.approx(0.9))]
    assert ap.state() in (1, 2)
    assert ap.feed(np.zeros(16384, np.int16)) == 1004          # the ring keeps one slot free
    assert ap.feed(np.zeros(16383, np.int16)) == 0 and ap.drain() == 0
    ap.close()


def test_wake_word_gate_and_partial_frames_do_not_spin(stub_lib):
    """1600-sample chunks leave 64 samples (less than a wake-word frame) in the ring: the worker must sleep on them, not spin with the
    mutex held (a regression this test pins: feed() starved forever)."""
    ap = make(stub_lib, wake_word="porcupine.pv")
    assert ap.state() == 1
    for chunk in np.split(SIG[:16000], 10):
        assert ap.feed(chunk) == 0
    assert ap.drain() == 0 and ap.vad_events == [] and ap.transcriptions == []
    assert ap.wake() == 0 and ap.state() == 2 and ap.wake() == 1002
    for chunk in np.split(SIG, 25):
        assert ap.feed(chunk) == 0 and ap.drain() == 0
    assert ap.vad_events == [0, 1] and len(ap.transcriptions) == 1 and ap.transcriptions[0][1]
    assert ap.state() == 1                                     # back to waiting for the wake word after the final transcription
    ap.close()


def test_tts_priority_queue_and_interruption(stub_lib):
    ap = make(stub_lib)
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
    assert spoken == [b"first"] and ap.state() == 4
    for text, pr in (("n1", NORMAL), ("low", LOW), ("h1", HIGH), ("n2", NORMAL), ("crit", CRITICAL), ("h2", HIGH)):
        assert ap.say(text, pr) == 0
    assert ap.interrupts >= 1
    gate.set()
    assert ap.drain() == 0
    assert spoken == [b"first", b"crit", b"h1", b"h2", b"n1", b"n2", b"low"]
    assert sum(1 for a, _ in ap.tts_audio if a[0] == 2) == 6   # the interrupted request's second chunk was dropped
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
    assert ap.say("fire", CRITICAL) == 0 and ap.interrupts == n_int      # a HIGH request is never interrupted
    for i in range(15):
        assert ap.say("x%d" % i, LOW) == (0 if i < 14 else 1004)         # 16 entries at most
    gate.set()
    assert ap.drain() == 0 and spoken[:2] == [b"urgent", b"fire"] and len(spoken) == 16
    ap.close()


def test_reference_listening_timeout_replay(stub_lib):
    """tests/tk_audio_pipeline_full_test.c:127-170 (test_listening_timeout): a wake-word pipeline fed 60 chunks of 1600 zero samples (6 s of
    silence) stays in AWAITING_WAKE_WORD; nothing is reported."""
    ap = make(stub_lib, wake_word="porcupine.pv")
    assert ap.state() == 1                                      # TK_PIPELINE_STATE_AWAITING_WAKE_WORD
    silence = np.zeros(1600, np.int16)
    for _ in range(60):
        assert ap.feed(silence) == 0
        time.sleep(0.002)                                       # the reference sleeps 100 ms per chunk; the state machine has no clock to wait for here
    assert ap.drain() == 0 and ap.state() == 1
    assert ap.vad_events == [] and ap.transcriptions == []
    ap.close()


def test_reference_tts_interruption_replay(stub_lib):
    """tests/tk_audio_pipeline_full_test.c:172-214 (test_tts_interruption): a LOW request in progress, a CRITICAL one 50 ms later ->
    on_tts_interrupt fires; the interrupted request's remaining audio is dropped and the critical text is spoken next."""
    ap = make(stub_lib)
    spoken = []

    def synth(text, emit):                                      # stands in for Piper: a long message takes a while, in chunks
        spoken.append(text)
        for _ in range(40 if len(text) > 20 else 1):
            emit(np.full(160, 3, np.int16))
            time.sleep(0.01)
        return 0

    ap.set_synthesizer(synth)
    assert ap.say("This is a long, low-priority message that should be interrupted.", LOW) == 0
    time.sleep(0.05)
    assert ap.say("Alert!", CRITICAL) == 0
    assert ap.drain() == 0
    assert ap.interrupts >= 1
    assert spoken == [b"This is a long, low-priority message that should be interrupted.", b"Alert!"]
    assert len(ap.tts_audio) < 41                               # the long message did not get all of its 40 chunks out
    ap.close()


def test_failed_asr_pass_drops_the_segment_instead_of_overrunning(stub_lib):
    """when the ASR engine fails at the 30 s limit (GPU error, or a segment longer than the engine's buffer at sample rates
    above 16 kHz) the segment must be dropped — the next chunk is copied to the START of the buffer, never past its end — and the
    engine is reset.  40 s of speech with every ASR call failing: no call ever sees more than the 30 s buffer; then the engine
    recovers and the next segment is transcribed from its own start."""
    ap = make(stub_lib)
    calls, resets, max_n = C.c_int(), C.c_int(), C.c_size_t()
    stub_lib.stub_asr_stats(C.byref(calls), C.byref(resets), C.byref(max_n))
    calls0, resets0 = calls.value, resets.value
    stub_lib.stub_asr_set_fail(1)
    loud = np.full(8000, 5000, np.int16)
    try:
        for _ in range(80):                                   # 40 s > the 30 s segment buffer
            assert ap.feed(loud) == 0 and ap.drain(5000) == 0
        stub_lib.stub_asr_stats(C.byref(calls), C.byref(resets), C.byref(max_n))
        assert calls.value - calls0 >= 1 and resets.value - resets0 == calls.value - calls0
        assert max_n.value <= 30 * 16000
        assert ap.force_end() != 0                            # the pending (failing) final pass reports the engine's error ...
        assert ap.force_end() == 0                            # ... and consumed the segment: nothing left to fail on
    finally:
        stub_lib.stub_asr_set_fail(0)
    assert ap.transcriptions == []
    ap.close()
