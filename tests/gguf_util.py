"""Minimal GGUF v3 writer for tests (public GGUF specification; test infrastructure only)."""
import struct

import numpy as np

GGUF_U32, GGUF_F32, GGUF_STRING, GGUF_ARRAY, GGUF_I32 = 4, 6, 8, 9, 5


def _s(b):
    b = b.encode() if isinstance(b, str) else b
    return struct.pack("<Q", len(b)) + b


def write_gguf(path, kv, tensors, align=32):
    """kv: list of (key, type, value); tensors: list of (name, dims(ne0 first), ggml_type, bytes)"""
    out = bytearray(b"GGUF" + struct.pack("<IQQ", 3, len(tensors), len(kv)))
    for key, t, v in kv:
        out += _s(key) + struct.pack("<I", t)
        if t == GGUF_U32:
            out += struct.pack("<I", v)
        elif t == GGUF_I32:
            out += struct.pack("<i", v)
        elif t == GGUF_F32:
            out += struct.pack("<f", v)
        elif t == GGUF_STRING:
            out += _s(v)
        elif t == GGUF_ARRAY:
            et, items = v
            out += struct.pack("<IQ", et, len(items))
            if et == GGUF_STRING:
                for it in items:
                    out += _s(it)
            elif et == GGUF_F32:
                out += np.asarray(items, np.float32).tobytes()
            elif et == GGUF_I32:
                out += np.asarray(items, np.int32).tobytes()
    off = 0
    offs = []
    for name, dims, typ, data in tensors:
        offs.append(off)
        off += (len(data) + align - 1) // align * align
    for (name, dims, typ, data), o in zip(tensors, offs):
        out += _s(name) + struct.pack("<I", len(dims)) + b"".join(struct.pack("<Q", d) for d in dims) + struct.pack("<IQ", typ, o)
    out += b"\0" * ((-len(out)) % align)
    for name, dims, typ, data in tensors:
        out += bytes(data) + b"\0" * ((-len(data)) % align)
    open(path, "wb").write(bytes(out))


def test_vocab(vocab):
    toks = ["<unk>", "<s>", "</s>"] + ["<0x%02X>" % b for b in range(256)] + ["▁", "▁he", "ll", "llo", "▁hello", "he", "l", "o", "h", "e", "▁w", "or", "▁wor", "ld", "▁world", "w", "r", "d"]
    toks += ["tok%d" % i for i in range(len(toks), vocab)]
    return toks[:vocab]


def expected_piece(vocab, tid):
    if tid < 3:
        return b""
    if tid < 259:
        return bytes([tid - 3])
    return test_vocab(vocab)[tid].replace("▁", " ").encode()


def write_llama_gguf(path, orc, cfg, with_vocab=True, drop=()):
    """serialise an OracleLlm (tests/oracle_lib.py) as a llama-architecture GGUF; `drop` = tensor names to leave out (damaged-file tests)"""
    import oracle_lib as O
    names = ["attn_norm", "attn_q", "attn_k", "attn_v", "attn_output", "ffn_norm", "ffn_gate", "ffn_up", "ffn_down"]
    kv = [("general.architecture", GGUF_STRING, "llama"), ("llama.block_count", GGUF_U32, cfg.n_layer),
          ("llama.embedding_length", GGUF_U32, cfg.d_model), ("llama.feed_forward_length", GGUF_U32, cfg.d_ff),
          ("llama.attention.head_count", GGUF_U32, cfg.n_head), ("llama.attention.head_count_kv", GGUF_U32, cfg.n_kv_head),
          ("llama.rope.dimension_count", GGUF_U32, cfg.head_dim), ("llama.context_length", GGUF_U32, 4096),
          ("llama.attention.layer_norm_rms_epsilon", GGUF_F32, cfg.rms_eps), ("llama.rope.freq_base", GGUF_F32, cfg.rope_theta)]
    if with_vocab:
        toks = ["<unk>", "<s>", "</s>"] + ["<0x%02X>" % b for b in range(256)] + ["▁", "▁he", "ll", "llo", "▁hello", "he", "l", "o", "h", "e", "▁w", "or", "▁wor", "ld", "▁world", "w", "r", "d"]
        toks += ["tok%d" % i for i in range(len(toks), cfg.vocab)]
        scores = [0.0] * 259 + [-1.0, -2.0, -3.5, -3.0, -1.5, -4.0, -9.0, -9.0, -9.0, -9.0, -5.0, -6.0, -2.2, -7.0, -2.5, -9.0, -9.0, -9.0]
        scores += [-100.0] * (cfg.vocab - len(scores))
        types = [2, 3, 3] + [6] * 256 + [1] * (cfg.vocab - 259)
        kv += [("tokenizer.ggml.model", GGUF_STRING, "llama"), ("tokenizer.ggml.tokens", GGUF_ARRAY, (GGUF_STRING, toks[:cfg.vocab])),
               ("tokenizer.ggml.scores", GGUF_ARRAY, (GGUF_F32, scores[:cfg.vocab])), ("tokenizer.ggml.token_type", GGUF_ARRAY, (GGUF_I32, types[:cfg.vocab])),
               ("tokenizer.ggml.bos_token_id", GGUF_U32, 1), ("tokenizer.ggml.eos_token_id", GGUF_U32, 2)]
    tensors = []

    def add(name, layer, which, rows, cols):
        t, buf = orc.get_tensor(layer, which)
        tensors.append((name, [cols, rows] if rows > 1 else [cols], t, buf.tobytes()))

    D, QD, KVD, FF, V = cfg.d_model, cfg.n_head * cfg.head_dim, cfg.n_kv_head * cfg.head_dim, cfg.d_ff, cfg.vocab
    add("token_embd.weight", -1, O.T_TOKEN_EMBD, V, D)
    add("output_norm.weight", -1, O.T_OUT_NORM, 1, D)
    add("output.weight", -1, O.T_OUTPUT, V, D)
    shapes = [(1, D), (QD, D), (KVD, D), (KVD, D), (D, QD), (1, D), (FF, D), (FF, D), (D, FF)]
    for l in range(cfg.n_layer):
        for w, (r, c) in enumerate(shapes):
            add(f"blk.{l}.{names[w]}.weight", l, w, r, c)
    write_gguf(path, kv, [t for t in tensors if t[0] not in drop])


LORA_NAMES = {1: "attn_q", 2: "attn_k", 3: "attn_v", 4: "attn_output", 6: "ffn_gate", 7: "ffn_up", 8: "ffn_down"}


def lora_base_name(layer, which):
    return "output.weight" if layer < 0 else "blk.%d.%s.weight" % (layer, LORA_NAMES[which])


def write_lora_ggla(path, r, alpha, factors, f16=False):
    """the legacy adapter container llama_model_apply_lora_from_file read (convert-lora-to-ggml.py's output).
    factors: {(layer, which): (A [r][k_in], B [n_out][r])}; loraA is stored transposed ([k_in][r], ne = {r, k_in}), loraB as is (ne = {r, n_out})"""
    out = bytearray(struct.pack("<IIii", 0x67676C61, 1, r, int(alpha)))
    dt = np.float16 if f16 else np.float32
    for (layer, which), (A, B) in factors.items():
        for kind, arr in (("loraA", np.ascontiguousarray(np.asarray(A, np.float32).T)), ("loraB", np.asarray(B, np.float32))):
            name = (lora_base_name(layer, which) + "." + kind).encode()
            out += struct.pack("<iii", 2, len(name), 1 if f16 else 0)
            out += struct.pack("<ii", arr.shape[1], arr.shape[0])           # ne[0] is the contiguous dimension
            out += name
            out += b"\0" * ((-len(out)) % 32)
            out += arr.astype(dt).tobytes()
    open(path, "wb").write(bytes(out))


def write_lora_gguf(path, alpha, factors, f16=False):
    """a GGUF adapter (convert_lora_to_gguf.py's output): <base>.lora_a ne = {k_in, r}, <base>.lora_b ne = {r, n_out}"""
    kv = [("general.architecture", GGUF_STRING, "llama"), ("general.type", GGUF_STRING, "adapter"), ("adapter.type", GGUF_STRING, "lora"),
          ("adapter.lora.alpha", GGUF_F32, float(alpha))]
    dt = np.float16 if f16 else np.float32
    tensors = []
    for (layer, which), (A, B) in factors.items():
        A = np.asarray(A, np.float32)
        B = np.asarray(B, np.float32)
        base = lora_base_name(layer, which)
        tensors.append((base + ".lora_a", [A.shape[1], A.shape[0]], 1 if f16 else 0, A.astype(dt).tobytes()))
        tensors.append((base + ".lora_b", [B.shape[1], B.shape[0]], 1 if f16 else 0, B.astype(dt).tobytes()))
    write_gguf(path, kv, tensors)
