"""LLM stream: tk_model_loader_* / tk_llm_runner_* and the batched tk_mi355x_llm_* extension."""
import ctypes as C

import numpy as np

from ._lib import check, lib


class LlmHParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_layer", "d_model", "n_head", "n_kv_head", "head_dim", "d_ff", "vocab")] + \
               [("rms_eps", C.c_float), ("rope_theta", C.c_float)] + \
               [(n, C.c_int32) for n in ("ks_qkv", "ks_o", "ks_gateup", "ks_down", "ks_out")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def lora_probe(path):
    """(rank, alpha, tensor pairs) of a LoRA adapter file; no GPU involved"""
    r, a, n = C.c_int32(0), C.c_float(0), C.c_int32(0)
    check(lib().tk_mi355x_lora_probe(path.encode(), C.byref(r), C.byref(a), C.byref(n)))
    return r.value, a.value, n.value


def attention_plan(nrows, n_head, n_kv_head, head_dim, max_ctx, fused=True, device=0, top_position=None):
    """the attention launch a pass takes on `device`: (kernel, query heads per workgroup, positions per slot / resident chunk, slots);
    kernel 0 = k_attention, 1 = k_attention_narrow, 2 = k_attention_prefill, 3 = the long-context decode form (tk_mi355x_attention_plan;
    with top_position — the pass's highest position — tk_mi355x_attention_plan_at: the session's per-pass choice)"""
    out = (C.c_int32 * 4)()
    if top_position is None:
        check(lib().tk_mi355x_attention_plan(device, nrows, n_head, n_kv_head, head_dim, max_ctx, 1 if fused else 0, out))
    else:
        check(lib().tk_mi355x_attention_plan_at(device, nrows, n_head, n_kv_head, head_dim, max_ctx, 1 if fused else 0, int(top_position), out))
    return tuple(out)


class Sampling(C.Structure):  # tk_mi355x_sampling_t
    _fields_ = [("temperature", C.c_float), ("top_p", C.c_float), ("min_p", C.c_float), ("top_k", C.c_int32), ("seed", C.c_uint64),
                ("counter", C.c_uint32), ("reserved", C.c_uint32)]


def MISTRAL_7B():
    return LlmHParams(32, 4096, 32, 8, 128, 14336, 32000, 1e-5, 10000.0, 0, 0, 0, 0, 1)


def TINY():
    return LlmHParams(2, 256, 8, 2, 64, 512, 512, 1e-5, 10000.0, 0, 0, 0, 0, 1)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class LlmModel:
    def __init__(self, hparams=None, device=0, gguf=None, lora=None):
        self.h = C.c_void_p()
        if gguf is not None:
            check(lib().tk_mi355x_llm_model_load_gguf_lora(C.byref(self.h), gguf.encode(), lora.encode() if lora else None, device))
        else:
            check(lib().tk_mi355x_llm_model_create(C.byref(self.h), C.byref(hparams), device))

    @property
    def hparams(self):
        hp = LlmHParams()
        lib().tk_mi355x_llm_model_get_hparams(self.h, C.byref(hp))
        return hp

    @property
    def weight_bytes(self):
        lib().tk_mi355x_llm_model_weight_bytes.restype = C.c_uint64
        return lib().tk_mi355x_llm_model_weight_bytes(self.h)

    def fill_synthetic(self, seed, f16=False):
        fn = lib().tk_mi355x_llm_model_fill_synthetic_f16 if f16 else lib().tk_mi355x_llm_model_fill_synthetic
        check(fn(self.h, C.c_uint64(seed)))
        return self

    def set_tensor(self, layer, which, ttype, data):
        data = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        check(lib().tk_mi355x_llm_model_set_tensor(self.h, layer, which, ttype, _p(data), C.c_size_t(data.size)))

    def set_lora(self, adapter_path):
        """the adapter merged into every matrix installed from now on (set_tensor, fill_synthetic); None = none"""
        check(lib().tk_mi355x_llm_model_set_lora(self.h, adapter_path.encode() if adapter_path else None))
        return self

    @property
    def lora_merged(self):
        return lib().tk_mi355x_llm_model_lora_merged(self.h)

    def close(self):
        if self.h:
            lib().tk_mi355x_llm_model_destroy(C.byref(self.h))
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LlmSession:
    def __init__(self, model, max_seq, max_ctx):
        self.model = model
        self.vocab = model.hparams.vocab
        self.h = C.c_void_p()
        check(lib().tk_mi355x_llm_session_create(C.byref(self.h), model.h, max_seq, max_ctx))

    def forward(self, seq, pos, tok, want_logits=True):
        seq = np.ascontiguousarray(seq, dtype=np.int32)
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        tok = np.ascontiguousarray(tok, dtype=np.int32)
        n = len(seq)
        logits = np.empty((n, self.vocab), dtype=np.float32) if want_logits else None
        am = np.empty(n, dtype=np.int32)
        check(lib().tk_mi355x_llm_forward(self.h, n, _p(seq), _p(pos), _p(tok), _p(logits), _p(am)))
        return logits, am

    def forward_sampled(self, seq, pos, tok, sampling, want_logits=True):
        """forward() with a sampling state per row: sampling = [(temperature, top_k, top_p, min_p, seed, counter), ...]; temperature <= 0
        takes the arg max for that row"""
        seq = np.ascontiguousarray(seq, dtype=np.int32)
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        tok = np.ascontiguousarray(tok, dtype=np.int32)
        n = len(seq)
        tab = (Sampling * n)()
        for r, (temp, top_k, top_p, min_p, seed, counter) in enumerate(sampling):
            tab[r] = Sampling(temp, top_p, min_p, top_k, seed, counter, 0)
        logits = np.empty((n, self.vocab), dtype=np.float32) if want_logits else None
        ids = np.empty(n, dtype=np.int32)
        check(lib().tk_mi355x_llm_forward_sampled(self.h, n, _p(seq), _p(pos), _p(tok), tab, _p(logits), _p(ids)))
        return logits, ids

    def forward_stage(self, seq, pos, layer0, layer1, tok=None, x_in=None, x_out=None, head=False, on_host=True):
        """layers [layer0, layer1) of one pass (pipeline stage).  x_in / x_out: float32 [n][d_model] numpy arrays (on_host) or raw
        device addresses (ints, e.g. torch_tensor.data_ptr()) on this session's GPU.  Returns arg max ids when head."""
        seq = np.ascontiguousarray(seq, dtype=np.int32)
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        n = len(seq)
        if tok is not None:
            tok = np.ascontiguousarray(tok, dtype=np.int32)
        am = np.empty(n, dtype=np.int32) if head else None
        ptr = (lambda a: _p(a)) if on_host else (lambda a: None if a is None else C.c_void_p(int(a)))
        check(lib().tk_mi355x_llm_forward_stage(self.h, n, _p(seq), _p(pos), _p(tok), ptr(x_in), ptr(x_out), 1 if on_host else 0,
                                                layer0, layer1, 1 if head else 0, _p(am)))
        return am

    def kv_write(self, layer, seq, pos0, k, v):
        """k, v: uint16 (f16 bits) [n_pos][n_kv_head][head_dim]"""
        k = np.ascontiguousarray(k, dtype=np.uint16)
        v = np.ascontiguousarray(v, dtype=np.uint16)
        assert k.shape == v.shape and k.ndim == 3
        check(lib().tk_mi355x_llm_session_kv_write(self.h, layer, seq, pos0, k.shape[0], _p(k), _p(v)))

    def kv_read(self, layer, seq, pos0, n_pos):
        hp = self.model.hparams
        k = np.empty((n_pos, hp.n_kv_head, hp.head_dim), np.uint16)
        v = np.empty_like(k)
        check(lib().tk_mi355x_llm_session_kv_read(self.h, layer, seq, pos0, n_pos, _p(k), _p(v)))
        return k, v

    def prefill(self, tokens):
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        nseq, n_prompt = tokens.shape
        first = np.empty(nseq, dtype=np.int32)
        check(lib().tk_mi355x_llm_prefill(self.h, nseq, n_prompt, _p(tokens), _p(first)))
        return first

    def decode(self, nrows, n_steps):
        out = np.empty((n_steps, lib().tk_mi355x_llm_max_rows()), dtype=np.int32)
        ms = C.c_float(0)
        check(lib().tk_mi355x_llm_decode(self.h, nrows, n_steps, _p(out), C.byref(ms)))
        return out[:, :nrows].copy(), ms.value

    def time_gemv(self, layer, which, nrows, iters):
        ms = C.c_float(0)
        nbytes = C.c_double(0)
        check(lib().tk_mi355x_llm_time_gemv(self.h, layer, which, nrows, iters, C.byref(ms), C.byref(nbytes)))
        return ms.value, nbytes.value

    def time_attention(self, nrows, ctx, iters):
        ms = C.c_float(0)
        nbytes = C.c_double(0)
        check(lib().tk_mi355x_llm_time_attention(self.h, nrows, ctx, iters, C.byref(ms), C.byref(nbytes)))
        return ms.value, nbytes.value

    def close(self):
        if self.h:
            lib().tk_mi355x_llm_session_destroy(C.byref(self.h))
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PipeHandle(C.Structure):
    _fields_ = [("ipc", C.c_uint8 * 64), ("bytes", C.c_uint64), ("device", C.c_int32), ("pid", C.c_int32)]

    def to_bytes(self):
        return bytes(memoryview(self))

    @classmethod
    def from_bytes(cls, b):
        return cls.from_buffer_copy(b)


class LlmPipe:
    """one stage of the layer-sharded LLM with the hand-off inside the library (tk_mi355x_pipe_*)"""

    def __init__(self, session, stage, n_stages, layer0, layer1, payload_f16=False):
        self.session, self.stage, self.n_stages = session, stage, n_stages
        self.h = C.c_void_p()
        self.handle = PipeHandle()
        check(lib().tk_mi355x_pipe_create(C.byref(self.h), session.h, stage, n_stages, layer0, layer1, 1 if payload_f16 else 0, C.byref(self.handle)))

    def connect(self, next_handle, prev_handle):
        check(lib().tk_mi355x_pipe_connect(self.h, C.byref(next_handle), C.byref(prev_handle)))

    def connect_local(self, nxt, prv):
        check(lib().tk_mi355x_pipe_connect_local(self.h, nxt.h, prv.h))

    @staticmethod
    def rccl_unique_id():
        """128 bytes naming a new RCCL communicator: made in ONE process, handed to every stage (tk_mi355x_pipe_rccl_unique_id)"""
        buf = (C.c_uint8 * 128)()
        check(lib().tk_mi355x_pipe_rccl_unique_id(buf))
        return bytes(buf)

    def connect_rccl(self, unique_id):
        """the collective transport (ncclSend / ncclRecv per boundary) instead of the mailboxes; needs one GPU per stage and FAILS otherwise"""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        check(lib().tk_mi355x_pipe_connect_rccl(self.h, buf))

    def enqueue(self, seq, pos, tok=None, head=False):
        seq = np.ascontiguousarray(seq, dtype=np.int32)
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        if tok is not None:
            tok = np.ascontiguousarray(tok, dtype=np.int32)
        check(lib().tk_mi355x_pipe_pass(self.h, len(seq), _p(seq), _p(pos), _p(tok), 1 if head else 0))

    def decode(self, nrows, n_steps):
        check(lib().tk_mi355x_pipe_decode(self.h, nrows, n_steps))

    def sync(self, nrows=0, n_steps=0):
        out = np.zeros((max(n_steps, 1), lib().tk_mi355x_llm_max_rows()), dtype=np.int32) if n_steps else None
        check(lib().tk_mi355x_pipe_sync(self.h, _p(out), n_steps))
        return out[:n_steps, :nrows].copy() if n_steps else None

    def close(self):
        if self.h:
            lib().tk_mi355x_pipe_destroy(C.byref(self.h))
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- reference surface (what trackie-core links against) ------------------------------------

class _Path(C.Structure):
    _fields_ = [("path_str", C.c_char_p), ("length", C.c_size_t), ("capacity", C.c_size_t)]


class _LoaderConfig(C.Structure):
    _fields_ = [("max_models", C.c_uint32), ("num_threads", C.c_uint32)]


class _LoadParams(C.Structure):
    _fields_ = [("model_path", C.POINTER(_Path)), ("model_type", C.c_uint32), ("force_reload", C.c_bool),
                ("gpu_layers", C.c_uint32), ("cpu_threads", C.c_uint32), ("use_mmap", C.c_bool), ("use_mlock", C.c_bool),
                ("numa", C.c_bool), ("seed", C.c_uint32), ("lora_adapter", C.c_char_p)]


class _LlmConfig(C.Structure):
    _fields_ = [("context_size", C.c_uint32), ("system_prompt", C.c_char_p), ("random_seed", C.c_uint32)]


class ModelLoader:
    def __init__(self, max_models=4):
        self.h = C.c_void_p()
        cfg = _LoaderConfig(max_models, 1)
        check(lib().tk_model_loader_create(C.byref(self.h), C.byref(cfg)))

    def load(self, path, lora_adapter=None, force_reload=False):
        """tk_model_loader_load_model; lora_adapter: a "ggla" or GGUF adapter file merged into the weights at load"""
        lib().tk_path_create.restype = C.POINTER(_Path)
        p = lib().tk_path_create(path.encode())
        params = _LoadParams(p, 1, bool(force_reload), 99, 1, True, False, False, 0, lora_adapter.encode() if lora_adapter else None)
        handle = C.c_void_p()
        try:
            check(lib().tk_model_loader_load_model(self.h, C.byref(params), C.byref(handle)))
        finally:
            lib().tk_path_destroy(C.byref(p))
        return handle

    def unload(self, handle):
        check(lib().tk_model_loader_unload_model(self.h, C.byref(handle)))

    @staticmethod
    def set_runner_slots(handle, slots):
        """sequences per shared decode session of the runners created on this model handle (before the first tk_llm_runner_create)"""
        check(lib().tk_mi355x_llm_model_set_runner_slots(handle, slots))

    @staticmethod
    def batch_stats(handle):
        """(passes, rows, widest pass) of the continuous-batching schedulers of this model"""
        p, r, m = C.c_uint64(0), C.c_uint64(0), C.c_int32(0)
        lib().tk_mi355x_llm_model_batch_stats(handle, C.byref(p), C.byref(r), C.byref(m))
        return p.value, r.value, m.value

    @staticmethod
    def run_ahead_wasted(handle):
        """run-ahead rows of this model's schedulers that no owner came back for"""
        f = lib().tk_mi355x_llm_model_run_ahead_wasted
        f.restype = C.c_uint64
        f.argtypes = [C.c_void_p]
        return int(f(handle))

    def close(self):
        if self.h:
            lib().tk_model_loader_destroy(C.byref(self.h))
            self.h = C.c_void_p()


class LlmRunner:
    """tk_llm_runner_* exactly as the reference's Rust GgufRunner drives it."""

    def __init__(self, model_handle, context_size=4096, random_seed=0):
        self.h = C.c_void_p()
        cfg = _LlmConfig(context_size, None, random_seed)
        check(lib().tk_llm_runner_create(C.byref(self.h), model_handle, C.byref(cfg)))
        lib().tk_llm_runner_generate_next_token.restype = C.c_void_p

    def prepare(self, prompt, use_tool_grammar=False):
        check(lib().tk_llm_runner_prepare_generation(self.h, prompt.encode(), use_tool_grammar))

    def set_sampling(self, temperature, top_k=40, top_p=0.95, min_p=0.05):
        """the reference's default stochastic chain (llama_sampling_default_params) instead of greedy; temperature 0 = greedy again"""
        check(lib().tk_mi355x_llm_runner_set_sampling(self.h, C.c_float(temperature), top_k, C.c_float(top_p), C.c_float(min_p)))

    def next_token(self):
        p = lib().tk_llm_runner_generate_next_token(self.h)
        if not p:
            return None
        if p == 1:
            return "<tool_call>"
        return C.string_at(p)

    def reset(self):
        check(lib().tk_llm_runner_reset_context(self.h))

    def close(self):
        if self.h:
            lib().tk_llm_runner_destroy(C.byref(self.h))
            self.h = C.c_void_p()
