"""tk_cortex_* and the ROCm HAL surface over ctypes."""
import ctypes as C
import threading

import numpy as np

from ._lib import check, lib
from .vision import VideoFrame, make_frame


class _ModelPaths(C.Structure):
    _fields_ = [(n, C.c_char_p) for n in ("llm_model", "object_detection_model", "depth_estimation_model", "asr_model", "tts_model_dir",
                                          "vad_model", "tesseract_data_dir")]


class _CortexConfig(C.Structure):
    _fields_ = [("model_paths", _ModelPaths), ("gpu_device_id", C.c_int), ("main_loop_frequency_hz", C.c_float), ("user_language", C.c_char_p),
                ("user_data", C.c_void_p)]


_STATE_CB = C.CFUNCTYPE(None, C.c_int, C.c_void_p)
_TTS_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p)


class _Callbacks(C.Structure):
    _fields_ = [("on_state_change", _STATE_CB), ("on_tts_audio_ready", _TTS_CB)]


class CortexStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("frames_processed", "frames_with_objects", "speech_segments", "llm_responses", "llm_tokens", "events_dropped",
                                          "responses_parsed", "actions_parsed")]


class Cortex:
    def __init__(self, llm=None, detector=None, asr=None, vad=None, device=0, depth=None):
        self.states = []
        self._scb = _STATE_CB(lambda s, u: self.states.append(s))
        self._tcb = _TTS_CB(lambda a, n, sr, u: None)
        enc = lambda s: s.encode() if s else None
        cfg = _CortexConfig(_ModelPaths(enc(llm), enc(detector), enc(depth), enc(asr), None, enc(vad), None), device, 10.0, b"en", None)
        self.h = C.c_void_p()
        check(lib().tk_cortex_create(C.byref(self.h), C.byref(cfg), _Callbacks(self._scb, self._tcb)))
        self._thread = None

    def start(self):
        self._thread = threading.Thread(target=lambda: lib().tk_cortex_run(self.h), daemon=True)
        self._thread.start()

    def stop(self):
        check(lib().tk_cortex_stop(self.h))
        if self._thread:
            self._thread.join(timeout=60)

    def inject_frame(self, arr):
        f, keep = make_frame(arr)
        return lib().tk_cortex_inject_video_frame(self.h, C.byref(f))

    def inject_audio(self, pcm):
        pcm = np.ascontiguousarray(pcm, np.int16)
        return lib().tk_cortex_inject_audio_frame(self.h, pcm.ctypes.data_as(C.c_void_p), C.c_size_t(pcm.size))

    def state(self):
        s = C.c_int(0)
        check(lib().tk_cortex_get_state(self.h, C.byref(s)))
        return s.value

    def stats(self):
        s = CortexStats()
        lib().tk_mi355x_cortex_get_stats(self.h, C.byref(s))
        return s

    def last_response(self):
        buf = C.create_string_buffer(4096)
        lib().tk_mi355x_cortex_last_response.restype = C.c_size_t
        lib().tk_mi355x_cortex_last_response(self.h, buf, 4096)
        return buf.value

    def last_prompt(self):
        buf = C.create_string_buffer(16384)
        lib().tk_mi355x_cortex_last_prompt.restype = C.c_size_t
        lib().tk_mi355x_cortex_last_prompt(self.h, buf, 16384)
        return buf.value

    def set_max_tokens(self, n):
        lib().tk_mi355x_cortex_set_max_response_tokens(self.h, n)

    def close(self):
        if self.h:
            lib().tk_cortex_destroy(C.byref(self.h))
            self.h = C.c_void_p()


class Float3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class PreprocessParams(C.Structure):
    _fields_ = [("d_input_image", C.c_void_p), ("input_width", C.c_uint32), ("input_height", C.c_uint32), ("input_stride_bytes", C.c_uint32),
                ("d_output_tensor", C.c_void_p), ("output_width", C.c_uint32), ("output_height", C.c_uint32), ("mean", Float3), ("std_dev", Float3),
                ("scale", C.c_float)]


class DepthPostParams(C.Structure):
    _fields_ = [("d_raw_depth_map", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("d_metric_depth_map", C.c_void_p),
                ("scale", C.c_float), ("shift", C.c_float)]


class DepthToPointsParams(C.Structure):
    _fields_ = [("d_metric_depth_map", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("d_point_cloud", C.c_void_p),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]


class _DispatchConfig(C.Structure):
    _fields_ = [("device_id", C.c_int)]


class RocmDispatcher:
    """tk_rocm_dispatch_*: opaque device buffers, async copies on dedicated streams, kernel launch wrappers."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        cfg = _DispatchConfig(device)
        check(lib().tk_rocm_dispatch_create(C.byref(self.h), C.byref(cfg)))
        lib().tk_rocm_dispatch_buffer_ptr.restype = C.c_void_p

    def malloc(self, nbytes):
        b = C.c_void_p()
        check(lib().tk_rocm_dispatch_malloc(self.h, C.byref(b), C.c_size_t(nbytes)))
        return b

    def free(self, b):
        lib().tk_rocm_dispatch_free(self.h, C.byref(b))

    def ptr(self, b):
        return lib().tk_rocm_dispatch_buffer_ptr(b)

    def upload(self, b, arr):
        arr = np.ascontiguousarray(arr)
        check(lib().tk_rocm_dispatch_upload_async(self.h, b, arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes)))
        self.sync()  # the numpy temporary must outlive the copy

    def download(self, b, shape, dtype):
        out = np.empty(shape, dtype)
        check(lib().tk_rocm_dispatch_download_async(self.h, out.ctypes.data_as(C.c_void_p), b, C.c_size_t(out.nbytes)))
        self.sync()
        return out

    def sync(self):
        check(lib().tk_rocm_dispatch_synchronize(self.h))

    def stream(self):
        s = C.c_void_p()
        check(lib().tk_rocm_dispatch_get_stream(self.h, C.byref(s)))
        return s

    def close(self):
        if self.h:
            lib().tk_rocm_dispatch_destroy(C.byref(self.h))
            self.h = C.c_void_p()
