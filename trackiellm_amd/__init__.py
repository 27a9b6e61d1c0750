"""trackiellm_amd — Python host-side mirror of the tk_* C-ABI of libtrackie_mi355x.so.

The product is the shared library (hand-written gfx950 HIP kernels behind TrackieLLM's own
C operator surface).  This package only binds it with ctypes so that tests, bench.py and
Python callers can drive exactly the entry points a C/Rust host would link against.
There is no Python or CPU fallback: importing works without a GPU (symbols can be inspected),
calling a compute entry without a gfx950 device returns a TK_ERROR_GPU_* code, and a missing
.so raises immediately.
"""
from ._lib import lib, TkError, check, LIB_PATH  # noqa: F401
from .llm import LlmHParams, LlmModel, LlmSession, LlmPipe, PipeHandle, ModelLoader, LlmRunner, MISTRAL_7B, TINY, attention_plan, lora_probe  # noqa: F401
from .vision import ObjectDetector, VisionPipeline, classify_attributes, preprocess, COCO80  # noqa: F401,E402
from .vision import DepthEstimator, depth_onnx_probe, fuse_data, fusion_reset, fusion_raw_distance  # noqa: F401,E402
from .audio import Asr, Vad, WhisperHP, WHISPER_TINY_EN, AudioPipeline  # noqa: F401,E402
from .cortex import Cortex, RocmDispatcher, PreprocessParams, DepthPostParams, DepthToPointsParams, Float3  # noqa: F401,E402
