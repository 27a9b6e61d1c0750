"""CPU: GGUF parsing through the C-ABI probe entry (no GPU needed) — metadata, geometry, error paths."""
import ctypes as C
import os

import numpy as np

import gguf_util
import oracle_lib as O


def probe(path):
    import trackiellm_amd as tk
    hp = tk.LlmHParams()
    nv = C.c_int32(0)
    rc = tk.lib().tk_mi355x_gguf_probe(path.encode(), C.byref(hp), C.byref(nv))
    return rc, hp, nv.value


def test_probe_reads_llama_hparams(tmp_path):
    cfg = O.tiny_config()
    orc = O.OracleLlm(cfg, seed=4)
    p = str(tmp_path / "tiny.gguf")
    gguf_util.write_llama_gguf(p, orc, cfg)
    rc, hp, nv = probe(p)
    assert rc == 0
    assert (hp.n_layer, hp.d_model, hp.n_head, hp.n_kv_head, hp.head_dim, hp.d_ff, hp.vocab) == (2, 256, 8, 2, 64, 512, 512)
    assert abs(hp.rms_eps - 1e-5) < 1e-12 and hp.rope_theta == 10000.0 and nv == 512
    assert hp.ks_out == 1 and hp.ks_qkv >= 1


def test_probe_error_paths(tmp_path):
    rc, _, _ = probe(str(tmp_path / "missing.gguf"))
    assert rc == 3001                                   # TK_ERROR_FILE_NOT_FOUND
    bad = tmp_path / "bad.gguf"
    bad.write_bytes(b"NOPE" + b"\0" * 64)
    rc, _, _ = probe(str(bad))
    assert rc == 3004                                   # TK_ERROR_FILE_CORRUPT
    trunc = tmp_path / "trunc.gguf"
    cfg = O.tiny_config()
    p = str(tmp_path / "ok.gguf")
    gguf_util.write_llama_gguf(p, O.OracleLlm(cfg, seed=4), cfg)
    trunc.write_bytes(open(p, "rb").read()[:4000])
    rc, _, _ = probe(str(trunc))
    assert rc in (3004, 4000)


def test_sentencepiece_bpe_tokenizer(tmp_path):
    import trackiellm_amd as tk
    cfg = O.tiny_config()
    p = str(tmp_path / "tiny.gguf")
    gguf_util.write_llama_gguf(p, O.OracleLlm(cfg, seed=4), cfg)
    ids = np.zeros(32, np.int32)

    def tok(text, bos=1):
        n = tk.lib().tk_mi355x_gguf_tokenize(p.encode(), text.encode(), bos, ids.ctypes.data_as(C.c_void_p), 32)
        return ids[:n].tolist()

    assert tok("hello world") == [1, 263, 273]                     # score-ordered bigram merges: [bos, "▁hello", "▁world"]
    assert tok("hello", 0) == [263]
    assert tok("hé") == [1, 259, 267, 3 + 0xC3, 3 + 0xA9]          # "▁", "h", then <0xXX> byte fallback for the unknown character
    assert tok("") == [1] and tok("", 0) == []                     # llama.cpp: empty text has no fragment to prefix — [bos] alone (ABI_NOTES.md)


def test_tokenizer_reproduces_the_sentencepiece_library(tmp_path):
    """independent pin: a BPE model trained by the `sentencepiece` library with the Llama / Mistral tokenizer settings
    (tests/golden/make_spm_golden.py), its vocabulary stored the way llama.cpp's converter writes it into a GGUF, and the LIBRARY's ids of
    56 strings — empty, leading / trailing / repeated spaces, multi-byte scripts, emoji, control bytes, byte fallback, the U+2581 marker
    itself — as expected outputs of tk_mi355x_gguf_tokenize."""
    import json
    import trackiellm_amd as tk
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "spm_bpe_tiny.json"), encoding="utf-8"))
    G = gguf_util
    kv = [("general.architecture", G.GGUF_STRING, "llama"), ("tokenizer.ggml.model", G.GGUF_STRING, "llama"),
          ("tokenizer.ggml.tokens", G.GGUF_ARRAY, (G.GGUF_STRING, fx["tokens"])), ("tokenizer.ggml.scores", G.GGUF_ARRAY, (G.GGUF_F32, fx["scores"])),
          ("tokenizer.ggml.token_type", G.GGUF_ARRAY, (G.GGUF_I32, fx["types"])), ("tokenizer.ggml.bos_token_id", G.GGUF_U32, fx["bos_id"]),
          ("tokenizer.ggml.eos_token_id", G.GGUF_U32, fx["eos_id"])]
    p = str(tmp_path / "vocab_only.gguf")                          # a vocabulary-only file, like llama.cpp's ggml-vocab-*.gguf
    G.write_gguf(p, kv, [])
    ids = np.zeros(256, np.int32)
    assert len(fx["cases"]) >= 50
    for c in fx["cases"]:
        for bos in (0, 1):
            n = tk.lib().tk_mi355x_gguf_tokenize(p.encode(), c["text"].encode(), bos, ids.ctypes.data_as(C.c_void_p), 256)
            assert ids[:n].tolist() == [fx["bos_id"]] * bos + c["ids"], (c["text"], ids[:max(n, 0)].tolist(), c["ids"])


def test_crafted_headers_are_rejected_not_crashed(tmp_path):
    """64-bit fields of an untrusted file: a string-array count the file cannot hold, dims whose product wraps, an offset that wraps
    the bounds test — each must come back as TK_ERROR_FILE_CORRUPT (3004), never an exception through the C-ABI or a wild pointer."""
    import struct
    G = gguf_util
    kv = [("general.architecture", G.GGUF_STRING, "llama"), ("llama.block_count", G.GGUF_U32, 1), ("llama.embedding_length", G.GGUF_U32, 256),
          ("llama.attention.head_count", G.GGUF_U32, 4), ("llama.feed_forward_length", G.GGUF_U32, 512)]
    lacking = tmp_path / "lacking.gguf"   # no feed_forward_length: a load error, not a division by zero in the K-split planner
    G.write_gguf(str(lacking), kv[:-1], [("token_embd.weight", [256, 512], 0, np.zeros(256 * 512, np.float32).tobytes())])
    assert probe(str(lacking))[0] == 4000
    good = tmp_path / "good.gguf"
    G.write_gguf(str(good), kv, [("token_embd.weight", [256, 512], 0, np.zeros(256 * 512, np.float32).tobytes())])
    assert probe(str(good))[0] == 0
    raw = bytearray(good.read_bytes())

    # (1) tokenizer string array whose count is 2^61: reserve() would throw / exhaust memory
    huge = bytearray(b"GGUF" + struct.pack("<IQQ", 3, 0, 1))
    huge += G._s("tokenizer.ggml.tokens") + struct.pack("<IIQ", G.GGUF_ARRAY, G.GGUF_STRING, 1 << 61)
    p1 = tmp_path / "huge_array.gguf"
    p1.write_bytes(bytes(huge) + b"\0" * 64)
    assert probe(str(p1))[0] == 3004

    # locate the tensor directory entry of the good file: name, n_dims, dims..., type, offset
    name = G._s("token_embd.weight")
    at = raw.index(name) + len(name)
    assert struct.unpack_from("<I", raw, at)[0] == 2
    dims_at, off_at = at + 4, at + 4 + 16 + 4

    # (2) dims whose product wraps to a small number (2^63 * 2 * ... ): n *= d overflow
    b2 = bytearray(raw)
    struct.pack_into("<QQ", b2, dims_at, 1 << 63, 4)
    p2 = tmp_path / "dims_wrap.gguf"
    p2.write_bytes(bytes(b2))
    assert probe(str(p2))[0] == 3004

    # (3) offset close to 2^64: data0 + offset + nbytes wraps past the bounds test
    b3 = bytearray(raw)
    struct.pack_into("<Q", b3, off_at, (1 << 64) - 4096)
    p3 = tmp_path / "offset_wrap.gguf"
    p3.write_bytes(bytes(b3))
    assert probe(str(p3))[0] == 3004

    # (4) dims that do not wrap but describe more data than the file holds
    b4 = bytearray(raw)
    struct.pack_into("<QQ", b4, dims_at, 1 << 20, 1 << 20)
    p4 = tmp_path / "too_big.gguf"
    p4.write_bytes(bytes(b4))
    assert probe(str(p4))[0] == 3004
