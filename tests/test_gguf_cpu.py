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
    assert tok("") == [1, 259]                                     # llama adds the space prefix even to empty text
