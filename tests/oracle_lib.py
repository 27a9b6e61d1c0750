"""ctypes binding of oracle/liboracle.so — test infrastructure only (never imported by the product)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

TYPE_F32, TYPE_F16, TYPE_Q4_K, TYPE_Q6_K = 0, 1, 12, 14
BLOCK_BYTES = {TYPE_F32: 4, TYPE_F16: 2, TYPE_Q4_K: 144, TYPE_Q6_K: 210}
BLOCK_ELEMS = {TYPE_F32: 1, TYPE_F16: 1, TYPE_Q4_K: 256, TYPE_Q6_K: 256}

T_TOKEN_EMBD, T_OUT_NORM, T_OUTPUT = 0, 1, 2
L_ATTN_NORM, L_Q, L_K, L_V, L_O, L_FFN_NORM, L_GATE, L_UP, L_DOWN = range(9)


class LlmConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("n_layer", "d_model", "n_head", "n_kv_head", "head_dim", "d_ff", "vocab", "max_ctx", "max_seq")] + \
               [("rms_eps", C.c_float), ("rope_theta", C.c_float)] + \
               [(n, C.c_int32) for n in ("ks_qkv", "ks_o", "ks_gateup", "ks_down", "ks_out")]


def tiny_config(**kw):
    cfg = dict(n_layer=2, d_model=256, n_head=8, n_kv_head=2, head_dim=64, d_ff=512, vocab=512, max_ctx=64,
               max_seq=4, rms_eps=1e-5, rope_theta=10000.0, ks_qkv=1, ks_o=1, ks_gateup=1, ks_down=1, ks_out=1)
    cfg.update(kw)
    return LlmConfig(**cfg)


def mistral7b_config(**kw):
    cfg = dict(n_layer=32, d_model=4096, n_head=32, n_kv_head=8, head_dim=128, d_ff=14336, vocab=32000,
               max_ctx=256, max_seq=16, rms_eps=1e-5, rope_theta=10000.0,
               ks_qkv=4, ks_o=4, ks_gateup=1, ks_down=7, ks_out=1)
    cfg.update(kw)
    return LlmConfig(**cfg)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.orc_llm_create.restype = C.c_void_p
        L.orc_llm_create.argtypes = [C.POINTER(LlmConfig)]
        L.orc_llm_destroy.argtypes = [C.c_void_p]
        L.orc_llm_synth.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_llm_synth_f16.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_llm_reset.argtypes = [C.c_void_p]
        L.orc_llm_set_tensor.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64]
        L.orc_llm_get_tensor.restype = C.c_int64
        L.orc_llm_get_tensor.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_int64]
        L.orc_llm_tensor_type.argtypes = [C.POINTER(LlmConfig), C.c_int, C.c_int]
        L.orc_llm_forward.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.orc_sample_row.restype = C.c_int32
        L.orc_sample_row.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_int32, C.c_float, C.c_float, C.c_uint64, C.c_uint32]
        L.orc_llm_kv_write.argtypes = [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 2
        L.orc_llm_kv_read.argtypes = [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 2
        L.orc_rmsnorm.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]
        L.orc_q8k_quantize.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_gemv_q8.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int] + [C.c_void_p] * 4
        L.orc_dequant_row.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.orc_quantize_rows.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
        for f in ("orc_expf", "orc_logf", "orc_tanhf", "orc_geluf", "orc_siluf", "orc_sqrtf"):
            getattr(L, f).restype = C.c_float
            getattr(L, f).argtypes = [C.c_float]
        L.orc_f32_to_f16.restype = C.c_uint16
        L.orc_f32_to_f16.argtypes = [C.c_float]
        L.orc_f16_to_f32.restype = C.c_float
        L.orc_f16_to_f32.argtypes = [C.c_uint16]
        _LIB = L
    return _LIB


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleLlm:
    def __init__(self, cfg, seed=None, f16=False):
        self.cfg = cfg
        self.h = lib().orc_llm_create(C.byref(cfg))
        if seed is not None:
            (lib().orc_llm_synth_f16 if f16 else lib().orc_llm_synth)(self.h, seed)

    def close(self):
        if self.h:
            lib().orc_llm_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def reset(self):
        lib().orc_llm_reset(self.h)

    def get_tensor(self, layer, which):
        t = C.c_int(0)
        n = lib().orc_llm_get_tensor(self.h, layer, which, C.byref(t), None, 0)
        buf = np.empty(n, dtype=np.uint8)
        lib().orc_llm_get_tensor(self.h, layer, which, C.byref(t), ptr(buf), n)
        return t.value, buf

    def set_tensor(self, layer, which, ttype, data):
        data = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        rc = lib().orc_llm_set_tensor(self.h, layer, which, ttype, ptr(data), data.size)
        assert rc == 0, "tensor size mismatch"

    def apply_lora(self, layer, which, A, B, scale):
        """W += scale (B A): A [r][k_in], B [n_out][r]; the tensor is quantised back to its own type"""
        A = np.ascontiguousarray(A, np.float32)
        B = np.ascontiguousarray(B, np.float32)
        L = lib()
        L.orc_llm_apply_lora.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float]
        rc = L.orc_llm_apply_lora(self.h, layer, which, ptr(A), ptr(B), A.shape[0], scale)
        assert rc == 0, "tensor cannot take an adapter"

    def dequant(self, layer, which, rows, cols):
        t, buf = self.get_tensor(layer, which)
        out = np.empty((rows, cols), dtype=np.float32)
        row = np.empty(cols, dtype=np.float32)
        for r in range(rows):
            lib().orc_dequant_row(t, ptr(buf), cols, r, ptr(row))
            out[r] = row
        return out

    def kv_write(self, layer, seq, pos0, k, v):
        """k, v: uint16 (f16 bits) [n_pos][n_kv_head][head_dim]"""
        k = np.ascontiguousarray(k, dtype=np.uint16)
        v = np.ascontiguousarray(v, dtype=np.uint16)
        assert k.shape == v.shape and k.shape[1:] == (self.cfg.n_kv_head, self.cfg.head_dim) and pos0 + k.shape[0] <= self.cfg.max_ctx
        lib().orc_llm_kv_write(self.h, layer, seq, pos0, k.shape[0], ptr(k), ptr(v))

    def kv_read(self, layer, seq, pos0, n_pos):
        k = np.empty((n_pos, self.cfg.n_kv_head, self.cfg.head_dim), np.uint16)
        v = np.empty_like(k)
        lib().orc_llm_kv_read(self.h, layer, seq, pos0, n_pos, ptr(k), ptr(v))
        return k, v

    def forward(self, seq, pos, tok, want_logits=True):
        assert np.max(pos) < self.cfg.max_ctx and np.min(pos) >= 0 and np.max(seq) < self.cfg.max_seq and np.min(seq) >= 0, "row outside the oracle's cache"
        seq = np.ascontiguousarray(seq, dtype=np.int32)
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        tok = np.ascontiguousarray(tok, dtype=np.int32)
        n = len(seq)
        logits = np.empty((n, self.cfg.vocab), dtype=np.float32) if want_logits else None
        am = np.empty(n, dtype=np.int32)
        lib().orc_llm_forward(self.h, n, ptr(seq), ptr(pos), ptr(tok), ptr(logits) if want_logits else None, ptr(am))
        return logits, am


def sample_row(logits, temp, top_k, top_p, min_p, seed, counter, allow=None):
    """the canonical stochastic sampler (oracle/tk_oracle_llm.cpp: orc_sample_row); allow = uint32 bit words or None"""
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    if allow is not None:
        allow = np.ascontiguousarray(allow, dtype=np.uint32)
    return int(lib().orc_sample_row(ptr(logits), logits.size, ptr(allow) if allow is not None else None, temp, top_k, top_p, min_p, seed, counter))


def q8k_quantize(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = x.size
    q = np.empty(n, dtype=np.int8)
    d = np.empty(n // 256, dtype=np.float32)
    bs = np.empty(n // 32, dtype=np.int32)
    lib().orc_q8k_quantize(ptr(x), n, ptr(q), ptr(d), ptr(bs))
    return q, d, bs


def quantize_rows(ttype, x):
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    out = np.empty(x.size // 256 * BLOCK_BYTES[ttype], dtype=np.uint8)
    lib().orc_quantize_rows(ttype, ptr(x), x.size, ptr(out))
    return out


def dequant_rows(ttype, blocks, rows, cols):
    out = np.empty((rows, cols), dtype=np.float32)
    row = np.empty(cols, dtype=np.float32)
    for r in range(rows):
        lib().orc_dequant_row(ttype, ptr(blocks), cols, r, ptr(row))
        out[r] = row
    return out


def gemv_q8(ttype, blocks, rows, K, ks, x):
    q, d, bs = q8k_quantize(x)
    y = np.empty(rows, dtype=np.float32)
    lib().orc_gemv_q8(ttype, ptr(blocks), rows, K, ks, ptr(q), ptr(d), ptr(bs), ptr(y))
    return y


def rmsnorm(x, w, eps):
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    out = np.empty_like(x)
    lib().orc_rmsnorm(ptr(x), ptr(w), x.size, eps, ptr(out))
    return out


# ---------------------------------------------------------------- vision oracle bindings
class _Frame(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("stride", C.c_uint32), ("format", C.c_int), ("data", C.c_void_p)]


IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], np.float32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], np.float32)


def preprocess(frame, tw, th, mean=IMAGENET_MEAN, std=IMAGENET_STD, nhwc=False, stride=None, bpp=3):
    """oracle restatement of the reference pre-processor; frame: uint8 [H][W][bpp] (or a padded buffer with stride)"""
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    h, w = frame.shape[0], frame.shape[1]
    stride = stride or frame.strides[0]
    out = np.empty((th, tw, 3) if nhwc else (3, th, tw), np.float32)
    L = lib()
    L.orc_preprocess.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                 C.c_void_p, C.c_void_p, C.c_int]
    rc = L.orc_preprocess(ptr(frame), w, h, stride, bpp, ptr(out), tw, th, ptr(np.ascontiguousarray(mean, np.float32)),
                          ptr(np.ascontiguousarray(std, np.float32)), int(nhwc))
    assert rc == 0
    return out


def ref_preprocess(frame, tw, th, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """the COMPILED reference function (oracle/_ref/libtkref_preprocess.so); frame uint8 [H][W][3] contiguous"""
    path = os.path.join(ROOT, "oracle", "_ref", "libtkref_preprocess.so")
    R = C.CDLL(path)
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    h, w = frame.shape[:2]
    f = _Frame(w, h, w * 3, 0, frame.ctypes.data)
    out = np.empty((3, th, tw), np.float32)
    R.tk_preprocessor_resize_and_normalize_to_chw.argtypes = [C.POINTER(_Frame), C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    rc = R.tk_preprocessor_resize_and_normalize_to_chw(C.byref(f), ptr(out), tw, th, ptr(np.ascontiguousarray(mean, np.float32)),
                                                       ptr(np.ascontiguousarray(std, np.float32)))
    assert rc == 0
    return out


class _RefFrame(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("stride", C.c_uint32), ("format", C.c_int), ("data", C.c_void_p)]


class _RefRect(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("w", C.c_int), ("h", C.c_int)]


def ref_attributes(frame, rect):
    """the COMPILED reference classifiers (oracle/_ref/libtkref_attr.so): frame uint8 [H][W][3] contiguous, rect (x, y, w, h) ->
    (colour name, door state)"""
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libtkref_attr.so"))
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    frame = np.ascontiguousarray(frame, np.uint8)
    f = _RefFrame(frame.shape[1], frame.shape[0], frame.shape[1] * 3, 0, frame.ctypes.data)
    r = _RefRect(*[int(v) for v in rect])
    out = []
    for fn in (L.tk_classify_dominant_color, L.tk_classify_door_state):
        fn.argtypes = [C.POINTER(_RefFrame), C.POINTER(_RefRect), C.POINTER(C.c_void_p)]
        p = C.c_void_p()
        assert fn(C.byref(f), C.byref(r), C.byref(p)) == 0
        out.append(C.string_at(p).decode())
        libc.free(p)
    return tuple(out)


def have_ref():
    return os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libtkref_preprocess.so"))


class OracleYolo:
    def __init__(self, nc=80, seed=5, cls_bias=-4.0):
        L = lib()
        L.orc_yolo_create.restype = C.c_void_p
        L.orc_yolo_create.argtypes = [C.c_int, C.c_uint64, C.c_float]
        L.orc_yolo_destroy.argtypes = [C.c_void_p]
        L.orc_yolo_layer_count.argtypes = [C.c_void_p]
        L.orc_yolo_layer_spec.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_yolo_get_layer.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_yolo_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_yolo_post.restype = C.c_int
        L.orc_yolo_post.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.nc = nc
        self.h = L.orc_yolo_create(nc, seed, cls_bias)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_yolo_destroy(self.h)
            self.h = None

    def layers(self):
        out = []
        for i in range(lib().orc_yolo_layer_count(self.h)):
            sp = np.zeros(5, np.int32)
            lib().orc_yolo_layer_spec(self.h, i, ptr(sp))
            cin, cout, k, s, act = [int(v) for v in sp]
            w = np.empty((cout, k, k, cin), np.float32)
            b = np.empty(cout, np.float32)
            lib().orc_yolo_get_layer(self.h, i, ptr(w), ptr(b))
            out.append(dict(cin=cin, cout=cout, k=k, s=s, act=act, w=w, b=b))
        return out

    @staticmethod
    def anchors(H, W):
        return (H // 8) * (W // 8) + (H // 16) * (W // 16) + (H // 32) * (W // 32)

    def forward(self, x):
        x = np.ascontiguousarray(x, np.float32)
        B, H, W, _ = x.shape
        raw = np.empty((B, self.anchors(H, W), 64 + self.nc), np.float32)
        lib().orc_yolo_forward(self.h, B, H, W, ptr(x), ptr(raw))
        return raw

    def post(self, raw1, H, W, conf, iou, cap=500):
        raw1 = np.ascontiguousarray(raw1, np.float32)
        boxes = np.zeros((cap, 5), np.float32)
        cls = np.zeros(cap, np.int32)
        anc = np.zeros(cap, np.int32)
        n = lib().orc_yolo_post(ptr(raw1), H, W, self.nc, conf, iou, ptr(boxes), ptr(cls), ptr(anc), cap)
        return boxes[:n], cls[:n], anc[:n]


def yolo_post_out(out, nc, conf, iou, cap=500):
    """decode + NMS of a YOLO-class graph's decoded output [4 + nc][anchors] (orc_yolo_post_out) -> (boxes5, classes, anchors)"""
    out = np.ascontiguousarray(out, np.float32)
    assert out.shape[0] == 4 + nc
    boxes = np.zeros((cap, 5), np.float32)
    cls = np.zeros(cap, np.int32)
    anc = np.zeros(cap, np.int32)
    L = lib()
    L.orc_yolo_post_out.restype = C.c_int
    L.orc_yolo_post_out.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    n = L.orc_yolo_post_out(ptr(out), nc, out.shape[1], conf, iou, ptr(boxes), ptr(cls), ptr(anc), cap)
    return boxes[:n], cls[:n], anc[:n]


def gemm(A, B, bias=None, residual=None, b_kn=False, act=0, alpha=1.0):
    A = np.ascontiguousarray(A, np.float32)
    B = np.ascontiguousarray(B, np.float32)
    M, K = A.shape
    N = B.shape[1] if b_kn else B.shape[0]
    Cm = np.empty((M, N), np.float32)
    L = lib()
    L.orc_gemm.argtypes = [C.c_void_p] * 5 + [C.c_int] * 9 + [C.c_float]
    bias = None if bias is None else np.ascontiguousarray(bias, np.float32)
    residual = None if residual is None else np.ascontiguousarray(residual, np.float32)
    L.orc_gemm(ptr(A), ptr(B), ptr(Cm), ptr(bias) if bias is not None else None, ptr(residual) if residual is not None else None,
               M, N, K, K, B.shape[1], N, N, int(b_kn), act, alpha)
    return Cm


# ---------------------------------------------------------------- audio oracle bindings
class WhisperHP(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
                                         "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer", "n_vocab")]


def whisper_tiny_test():
    return WhisperHP(80, 50, 64, 2, 2, 32, 64, 2, 2, 512)


def whisper_tiny_en():
    return WhisperHP(80, 1500, 384, 6, 4, 448, 384, 6, 4, 51864)


def whisper_prompt(hp):
    v = hp.n_vocab
    return np.array([min(50257, v - 3), min(50362, v - 1)], np.int32)


class OracleWhisper:
    def __init__(self, hp, seed=6):
        L = lib()
        L.orc_whisper_create.restype = C.c_void_p
        L.orc_whisper_create.argtypes = [C.POINTER(WhisperHP), C.c_uint64]
        L.orc_whisper_destroy.argtypes = [C.c_void_p]
        L.orc_whisper_tensor_count.argtypes = [C.c_void_p]
        L.orc_whisper_tensor_info.restype = C.c_int64
        L.orc_whisper_tensor_info.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.orc_whisper_get_tensor.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_whisper_transcribe.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4
        self.hp = hp
        self.h = L.orc_whisper_create(C.byref(hp), seed)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_whisper_destroy(self.h)
            self.h = None

    def tensors(self):
        out = {}
        for i in range(lib().orc_whisper_tensor_count(self.h)):
            name = C.create_string_buffer(128)
            r, c = C.c_int64(0), C.c_int64(0)
            lib().orc_whisper_tensor_info(self.h, i, name, 128, C.byref(r), C.byref(c))
            a = np.empty((r.value, c.value), np.float32)
            lib().orc_whisper_get_tensor(self.h, i, ptr(a))
            out[name.value.decode()] = a
        return out

    def set_tensors(self, tensors):
        """name -> [rows, cols] float32 arrays, any subset of tensors()"""
        L = lib()
        L.orc_whisper_set_tensor.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        for i in range(L.orc_whisper_tensor_count(self.h)):
            name = C.create_string_buffer(128)
            r, c = C.c_int64(0), C.c_int64(0)
            L.orc_whisper_tensor_info(self.h, i, name, 128, C.byref(r), C.byref(c))
            a = tensors.get(name.value.decode())
            if a is not None:
                a = np.ascontiguousarray(a, np.float32)
                assert a.shape == (r.value, c.value)
                L.orc_whisper_set_tensor(self.h, i, ptr(a))

    def transcribe(self, pcm, n_steps, want=("mel", "enc", "logits"), prompt=None):
        pcm = np.ascontiguousarray(pcm, np.int16)
        B, n = pcm.shape
        hp = self.hp
        prompt = whisper_prompt(hp) if prompt is None else np.ascontiguousarray(prompt, np.int32)
        toks = np.zeros((B, max(n_steps, 1)), np.int32)
        mel = np.empty((B, 2 * hp.n_audio_ctx, hp.n_mels), np.float32)
        enc = np.empty((B, hp.n_audio_ctx, hp.n_audio_state), np.float32)
        lg = np.empty((B, hp.n_vocab), np.float32)
        lib().orc_whisper_transcribe(self.h, B, ptr(pcm), n, ptr(prompt), len(prompt), n_steps, ptr(toks), ptr(mel), ptr(enc), ptr(lg))
        return toks[:, :n_steps], mel, enc, lg


    def transcribe_ref(self, pcm, lengths, n_steps, suppress, token_beg, token_eot, prompt, temperature=0.0, seed=0):
        """the decode under the reference's whisper.cpp parameters (orc_whisper_transcribe_ref): pcm [B][n] zero padded, lengths [B] ->
        (tokens [B][n_steps], logprobs [B][n_steps], result_len [B], status [B])"""
        pcm = np.ascontiguousarray(pcm, np.int16)
        B, n = pcm.shape
        lengths = np.ascontiguousarray(lengths, np.int32)
        suppress = np.ascontiguousarray(suppress, np.uint8)
        prompt = np.ascontiguousarray(prompt, np.int32)
        toks = np.zeros((B, n_steps), np.int32)
        lp = np.zeros((B, n_steps), np.float32)
        rl = np.zeros(B, np.int32)
        st = np.zeros(B, np.int32)
        L = lib()
        L.orc_whisper_transcribe_ref.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_void_p,
                                                 C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_whisper_transcribe_ref(self.h, B, ptr(pcm), n, ptr(lengths), ptr(prompt), len(prompt), n_steps, temperature, seed, ptr(suppress), token_beg, token_eot,
                                     ptr(toks), ptr(lp), ptr(rl), ptr(st))
        return toks, lp, rl, st

    def transcribe_policy(self, pcm, n_steps, temperature, seed, prompt=None):
        """forced decode with the token picked by temperature -> (tokens [B][n_steps], logprobs [B][n_steps])"""
        pcm = np.ascontiguousarray(pcm, np.int16)
        B, n = pcm.shape
        prompt = whisper_prompt(self.hp) if prompt is None else np.ascontiguousarray(prompt, np.int32)
        toks = np.zeros((B, n_steps), np.int32)
        lp = np.zeros((B, n_steps), np.float32)
        L = lib()
        L.orc_whisper_transcribe_policy.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_uint64,
                                                    C.c_void_p, C.c_void_p]
        L.orc_whisper_transcribe_policy(self.h, B, ptr(pcm), n, ptr(prompt), len(prompt), n_steps, temperature, seed, ptr(toks), ptr(lp))
        return toks, lp


def whisper_decode_failed(toks, lp, eot, entropy_thold=2.4, logprob_thold=-1.0):
    """whisper.cpp's acceptance test of one decode -> (failed, mean log-probability)"""
    toks = np.ascontiguousarray(toks, np.int32)
    lp = np.ascontiguousarray(lp, np.float32)
    avg = C.c_float(0)
    L = lib()
    L.orc_whisper_decode_failed.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int32, C.c_float, C.c_float, C.POINTER(C.c_float)]
    f = L.orc_whisper_decode_failed(ptr(toks), ptr(lp), toks.size, eot, entropy_thold, logprob_thold, C.byref(avg))
    return bool(f), avg.value


def vad_probabilities(seed, windows, hidden=64):
    windows = np.ascontiguousarray(windows, np.float32)
    n, w = windows.shape
    out = np.empty(n, np.float32)
    L = lib()
    L.orc_vad_probabilities.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.orc_vad_probabilities(seed, w, hidden, ptr(windows), n, ptr(out))
    return out


class VadState(C.Structure):
    _fields_ = [("threshold", C.c_float), ("min_silence_ms", C.c_float), ("min_speech_ms", C.c_float), ("active", C.c_int),
                ("triggered", C.c_int), ("prob", C.c_float), ("silence_ms", C.c_float), ("speech_ms", C.c_float), ("since_event_ms", C.c_float)]


def vad_run(probs, threshold=0.5, min_silence_ms=300.0, min_speech_ms=250.0, dt_ms=30.0):
    """reference state machine over a probability trace -> list of (index, event)"""
    L = lib()
    L.orc_vad_step.restype = C.c_int
    L.orc_vad_step.argtypes = [C.POINTER(VadState), C.c_float, C.c_float]
    s = VadState(threshold, min_silence_ms, min_speech_ms, 0, 0, 0, 0, 0, 0)
    ev = []
    for i, p in enumerate(probs):
        e = L.orc_vad_step(C.byref(s), float(p), dt_ms)
        if e >= 0:
            ev.append((i, e))
    return ev, s


def whisper_filter_pick(logits, suppress, state, beg, eot, tid0=50, temperature=0.0, seed=0, counter=0):
    """one sampled position of whisper.cpp's decoder under the reference's parameters (orc_whisper_filter_pick): logits [n_vocab], suppress [n_vocab]
    uint8, state int32[8] (updated in place) -> (token, logprob)"""
    logits = np.ascontiguousarray(logits, np.float32)
    suppress = np.ascontiguousarray(suppress, np.uint8)
    assert state.dtype == np.int32 and state.shape == (8,) and state.flags.c_contiguous
    lp = C.c_float(0)
    L = lib()
    L.orc_whisper_filter_pick.restype = C.c_int32
    L.orc_whisper_filter_pick.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_uint64, C.c_uint32, C.POINTER(C.c_float)]
    tok = L.orc_whisper_filter_pick(ptr(logits), len(logits), ptr(suppress), ptr(state), beg, eot, tid0, temperature, seed, counter, C.byref(lp))
    return int(tok), float(lp.value)
