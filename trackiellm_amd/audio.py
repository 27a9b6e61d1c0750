"""ASR + VAD streams: tk_asr_whisper_* / tk_vad_silero_* (+ batched extensions) over ctypes."""
import ctypes as C

import numpy as np

from ._lib import check, lib
from .llm import _Path


class WhisperHP(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
                                         "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer", "n_vocab")]


def WHISPER_TINY_EN():
    return WhisperHP(80, 1500, 384, 6, 4, 448, 384, 6, 4, 51864)


class _AsrConfig(C.Structure):
    _fields_ = [("model_path", C.POINTER(_Path)), ("language", C.c_char_p), ("translate_to_en", C.c_bool), ("sample_rate", C.c_uint32),
                ("user_data", C.c_void_p), ("n_threads", C.c_int), ("max_context", C.c_int), ("word_threshold", C.c_float)]


class AsrResult(C.Structure):
    _fields_ = [("text", C.c_char_p), ("text_length", C.c_size_t), ("confidence", C.c_float), ("is_partial", C.c_bool)]


class Asr:
    def __init__(self, model="synthetic://whisper-tiny.en?seed=6", hp=None, seed=6, device=0, max_batch=1, sample_rate=16000):
        self.h = C.c_void_p()
        if hp is not None:
            self.hp = hp
            check(lib().tk_mi355x_asr_create(C.byref(self.h), C.byref(hp), C.c_uint64(seed), device, max_batch))
        else:
            self.hp = WHISPER_TINY_EN()
            lib().tk_path_create.restype = C.POINTER(_Path)
            p = lib().tk_path_create(model.encode())
            cfg = _AsrConfig(p, b"en", False, sample_rate, None, 4, 16384, 0.01)
            try:
                check(lib().tk_asr_whisper_create(C.byref(self.h), C.byref(cfg)))
            finally:
                lib().tk_path_destroy(C.byref(p))
            check(lib().tk_mi355x_asr_get_hparams(self.h, C.byref(self.hp)))  # a ggml checkpoint brings its own geometry

    def process_audio(self, pcm, is_final):
        pcm = np.ascontiguousarray(pcm, np.int16)
        res = C.POINTER(AsrResult)()
        check(lib().tk_asr_whisper_process_audio(self.h, pcm.ctypes.data_as(C.c_void_p), C.c_size_t(pcm.size), bool(is_final), C.byref(res)))
        out = (res.contents.text, res.contents.text_length, res.contents.confidence, res.contents.is_partial)
        out = (out[0].decode() if out[0] else None,) + out[1:]
        lib().tk_asr_whisper_free_result(C.byref(res))
        return out

    def transcribe_tokens(self, pcm, n_steps, want_aux=True):
        pcm = np.ascontiguousarray(pcm, np.int16)
        B, n = pcm.shape
        hp = self.hp
        toks = np.zeros((B, n_steps), np.int32)
        mel = np.empty((B, 2 * hp.n_audio_ctx, hp.n_mels), np.float32) if want_aux else None
        enc = np.empty((B, hp.n_audio_ctx, hp.n_audio_state), np.float32) if want_aux else None
        lg = np.empty((B, hp.n_vocab), np.float32) if want_aux else None
        p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        check(lib().tk_mi355x_asr_transcribe_tokens(self.h, B, p(pcm), n, n_steps, p(toks), p(mel), p(enc), p(lg)))
        return toks, mel, enc, lg

    def transcribe_policy(self, pcm, n_steps, temperature=0.0, seed=0):
        """the forced decode with whisper.cpp's per-step bookkeeping -> (tokens [B][n_steps], logprobs [B][n_steps])"""
        pcm = np.ascontiguousarray(pcm, np.int16)
        B, n = pcm.shape
        toks = np.zeros((B, n_steps), np.int32)
        lp = np.zeros((B, n_steps), np.float32)
        check(lib().tk_mi355x_asr_transcribe_policy(self.h, B, pcm.ctypes.data_as(C.c_void_p), n, n_steps, C.c_float(temperature), C.c_uint64(seed),
                                                    toks.ctypes.data_as(C.c_void_p), lp.ctypes.data_as(C.c_void_p)))
        return toks, lp

    def set_decode_policy(self, enable=True, temperature_inc=0.2, entropy_thold=2.4, logprob_thold=-1.0, seed=0):
        """whisper.cpp's temperature fallback on final results, with the thresholds the reference sets (tk_asr_whisper.c:126-138)"""
        check(lib().tk_mi355x_asr_set_decode_policy(self.h, int(enable), C.c_float(temperature_inc), C.c_float(entropy_thold),
                                                    C.c_float(logprob_thold), C.c_uint64(seed)))

    def last_decode(self):
        """(temperature, mean log-probability, attempts) of the decode the last process_audio returned"""
        t, a, n = C.c_float(0), C.c_float(0), C.c_int32(0)
        lib().tk_mi355x_asr_last_decode(self.h, C.byref(t), C.byref(a), C.byref(n))
        return t.value, a.value, n.value

    def set_reference_decode(self, enable=True):
        """process_audio decodes under the reference's whisper.cpp parameters (logit filters, timestamps) — the default; False = the forced-greedy decode"""
        check(lib().tk_mi355x_asr_set_reference_decode(self.h, int(enable)))

    def set_fast_contraction(self, on=True):
        """opt-in: the long passes (log-mel, encoder) on the f16 matrix pipe with split operands (~1e-6 of scale off the exact chains)"""
        check(lib().tk_mi355x_asr_set_fast_contraction(self.h, 1 if on else 0))

    def transcribe_ref(self, pcm, n_steps, temperature=0.0, seed=0):
        """one utterance through the reference-parameter decode -> (tokens [n_steps], logprobs [n_steps], result_len, status)"""
        pcm = np.ascontiguousarray(pcm, np.int16).reshape(-1)
        toks = np.zeros(n_steps, np.int32)
        lp = np.zeros(n_steps, np.float32)
        rl, st = C.c_int32(0), C.c_int32(0)
        check(lib().tk_mi355x_asr_transcribe_ref(self.h, pcm.ctypes.data_as(C.c_void_p), pcm.size, n_steps, C.c_float(temperature), C.c_uint64(seed),
                                                 toks.ctypes.data_as(C.c_void_p), lp.ctypes.data_as(C.c_void_p), C.byref(rl), C.byref(st)))
        return toks, lp, rl.value, st.value

    def suppress_table(self):
        """(table uint8 [n_vocab] or None, token_beg, token_eot) of the reference-parameter decode"""
        beg, eot = C.c_int32(0), C.c_int32(0)
        n = lib().tk_mi355x_asr_suppress_table(self.h, None, 0, C.byref(beg), C.byref(eot))
        if n <= 0:
            return None, 0, 0
        tab = np.zeros(n, np.uint8)
        lib().tk_mi355x_asr_suppress_table(self.h, tab.ctypes.data_as(C.c_void_p), n, C.byref(beg), C.byref(eot))
        return tab, beg.value, eot.value

    def share_stats(self):
        """(live contexts, batched jobs, utterances, widest job) of the engine this context shares with the others opened on the same file"""
        v = [C.c_uint64(0) for _ in range(4)]
        lib().tk_mi355x_asr_share_stats(self.h, *[C.byref(x) for x in v])
        return tuple(int(x.value) for x in v)

    def set_language(self, lang):
        return lib().tk_asr_whisper_set_language(self.h, lang.encode() if lang is not None else None)

    def prompt_tokens(self):
        buf = (C.c_int32 * 8)()
        n = lib().tk_mi355x_asr_prompt_tokens(self.h, buf, 8)
        return None if n < 0 else list(buf[:n])

    def set_decode_steps(self, n):
        lib().tk_mi355x_asr_set_decode_steps(self.h, n)

    def reset(self):
        check(lib().tk_asr_whisper_reset(self.h))

    def close(self):
        if self.h:
            lib().tk_asr_whisper_destroy(C.byref(self.h))
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _VadConfig(C.Structure):
    _fields_ = [("model_path", C.POINTER(_Path)), ("sample_rate", C.c_uint32), ("user_data", C.c_void_p), ("threshold", C.c_float),
                ("min_silence_duration_ms", C.c_float), ("min_speech_duration_ms", C.c_float), ("speech_pad_ms", C.c_float)]


class VadStateC(C.Structure):
    _fields_ = [("is_speech_active", C.c_bool), ("speech_probability", C.c_float), ("silence_duration_ms", C.c_float),
                ("speech_duration_ms", C.c_float)]


_VAD_CB = C.CFUNCTYPE(None, C.c_int, C.c_void_p)


class Vad:
    def __init__(self, model="synthetic://vad?seed=7", sample_rate=16000, threshold=0.0, min_silence_ms=0.0, min_speech_ms=0.0):
        lib().tk_path_create.restype = C.POINTER(_Path)
        p = lib().tk_path_create(model.encode())
        cfg = _VadConfig(p, sample_rate, None, threshold, min_silence_ms, min_speech_ms, 30.0)
        self.h = C.c_void_p()
        try:
            check(lib().tk_vad_silero_create(C.byref(self.h), C.byref(cfg)))
        finally:
            lib().tk_path_destroy(C.byref(p))

    def probability(self, pcm):
        pcm = np.ascontiguousarray(pcm, np.int16)
        out = C.c_float(0)
        check(lib().tk_vad_silero_process_audio(self.h, pcm.ctypes.data_as(C.c_void_p), C.c_size_t(pcm.size), C.byref(out)))
        return out.value

    def process_with_events(self, pcm):
        pcm = np.ascontiguousarray(pcm, np.int16)
        events = []
        cb = _VAD_CB(lambda e, u: events.append(e))
        check(lib().tk_vad_silero_process_audio_with_events(self.h, pcm.ctypes.data_as(C.c_void_p), C.c_size_t(pcm.size), cb, None))
        return events

    def probabilities(self, windows):
        windows = np.ascontiguousarray(windows, np.float32)
        out = np.empty(windows.shape[0], np.float32)
        check(lib().tk_mi355x_vad_probabilities(self.h, windows.ctypes.data_as(C.c_void_p), windows.shape[0], out.ctypes.data_as(C.c_void_p)))
        return out

    def step(self, p):
        return lib().tk_mi355x_vad_step(self.h, C.c_float(p))

    def state(self):
        s = VadStateC()
        check(lib().tk_vad_silero_get_state(self.h, C.byref(s)))
        return s

    def reset(self):
        check(lib().tk_vad_silero_reset(self.h))

    def set_threshold(self, t):
        return lib().tk_vad_silero_set_threshold(self.h, C.c_float(t))

    def close(self):
        if self.h:
            lib().tk_vad_silero_destroy(C.byref(self.h))
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- audio pipeline state machine (tk_audio_pipeline_*) ----

class _AudioParams(C.Structure):
    _fields_ = [("sample_rate", C.c_uint32), ("channels", C.c_uint32)]


class _PipelineConfig(C.Structure):
    _fields_ = [("input_audio_params", _AudioParams), ("user_language", C.c_char_p), ("user_data", C.c_void_p), ("asr_model_path", C.POINTER(_Path)),
                ("vad_model_path", C.POINTER(_Path)), ("tts_model_path", C.POINTER(_Path)), ("tts_config_path", C.POINTER(_Path)),
                ("ww_model_path", C.POINTER(_Path)), ("ww_keyword_path", C.POINTER(_Path)), ("ww_sensitivity", C.c_float),
                ("sc_model_path", C.POINTER(_Path)), ("vad_silence_threshold_ms", C.c_float), ("vad_speech_probability_threshold", C.c_float)]


class Transcription(C.Structure):
    _fields_ = [("text", C.c_char_p), ("is_final", C.c_bool), ("confidence", C.c_float)]


_TRANS_CB = C.CFUNCTYPE(None, C.POINTER(Transcription), C.c_void_p)
_TTS_AUDIO_CB = C.CFUNCTYPE(None, C.POINTER(C.c_int16), C.c_size_t, C.c_uint32, C.c_void_p)
_TTS_INT_CB = C.CFUNCTYPE(None, C.c_void_p)
_AMBIENT_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
_EMIT_FN = C.CFUNCTYPE(None, C.POINTER(C.c_int16), C.c_size_t, C.c_uint32, C.c_void_p)
_SYNTH_FN = C.CFUNCTYPE(C.c_int, C.c_char_p, _EMIT_FN, C.c_void_p, C.c_void_p)


class _AudioCallbacks(C.Structure):
    _fields_ = [("on_vad_event", _VAD_CB), ("on_transcription", _TRANS_CB), ("on_tts_audio_ready", _TTS_AUDIO_CB), ("on_tts_interrupt", _TTS_INT_CB),
                ("on_ambient_sound_detected", _AMBIENT_CB)]


class AudioPipeline:
    """tk_audio_pipeline_*: ring -> (wake word) -> VAD -> ASR -> transcription callback; priority TTS queue in front of a pluggable synthesiser."""

    def __init__(self, asr="synthetic://whisper-tiny.en?seed=6", vad="synthetic://vad?seed=7", wake_word=None, threshold=0.8, silence_ms=500.0,
                 sample_rate=16000, library=None):
        self._lib = library if library is not None else lib()   # tests drive the host state machine against a stub-engine build
        self.vad_events, self.transcriptions, self.tts_audio, self.interrupts = [], [], [], 0
        self._cbs = _AudioCallbacks(
            _VAD_CB(lambda e, u: self.vad_events.append(e)),
            _TRANS_CB(lambda t, u: self.transcriptions.append((t.contents.text.decode(), t.contents.is_final, t.contents.confidence))),
            _TTS_AUDIO_CB(lambda a, n, sr, u: self.tts_audio.append((np.ctypeslib.as_array(a, (n,)).copy(), sr))),
            _TTS_INT_CB(lambda u: setattr(self, "interrupts", self.interrupts + 1)),
            _AMBIENT_CB(lambda r, u: None))
        self._lib.tk_path_create.restype = C.POINTER(_Path)
        paths = [self._lib.tk_path_create(s.encode()) if s else None for s in (asr, vad, wake_word)]
        cfg = _PipelineConfig(_AudioParams(sample_rate, 1), b"en", None, paths[0], paths[1], None, None, paths[2], None, 0.5, None, silence_ms, threshold)
        self.h = C.c_void_p()
        try:
            rc = self._lib.tk_audio_pipeline_create(C.byref(self.h), C.byref(cfg), self._cbs)
            if rc != 0:
                raise RuntimeError("tk_audio_pipeline_create failed: %d" % rc)
        finally:
            for p in paths:
                if p:
                    self._lib.tk_path_destroy(C.byref(p))
        self._synth = None

    def feed(self, pcm):
        pcm = np.ascontiguousarray(pcm, np.int16)
        return self._lib.tk_audio_pipeline_process_chunk(self.h, pcm.ctypes.data_as(C.c_void_p), C.c_size_t(pcm.size))

    def say(self, text, priority):
        return self._lib.tk_audio_pipeline_synthesize_text(self.h, text.encode(), priority)

    def set_synthesizer(self, fn):
        """fn(text: bytes, emit(pcm int16 array, sample_rate)) -> int error code, called on the pipeline's worker thread"""
        def tramp(text, emit, ctx, user):
            def e(pcm, sr=22050):
                pcm = np.ascontiguousarray(pcm, np.int16)
                emit(pcm.ctypes.data_as(C.POINTER(C.c_int16)), pcm.size, sr, ctx)
            return fn(text, e) or 0
        self._synth = _SYNTH_FN(tramp)
        assert self._lib.tk_mi355x_audio_pipeline_set_synthesizer(self.h, self._synth, None) == 0

    def wake(self):
        return self._lib.tk_mi355x_audio_pipeline_trigger_wake_word(self.h)

    def drain(self, timeout_ms=30000):
        return self._lib.tk_mi355x_audio_pipeline_drain(self.h, timeout_ms)

    def force_end(self):
        return self._lib.tk_audio_pipeline_force_transcription_end(self.h)

    def state(self):
        return self._lib.tk_audio_pipeline_get_state(self.h)

    def close(self):
        if self.h:
            self._lib.tk_audio_pipeline_destroy(C.byref(self.h))
            self.h = C.c_void_p()
