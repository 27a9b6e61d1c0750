"""whisper.cpp ggml checkpoint reader, parsing only (no GPU): header, filter bank, vocabulary, tensor directory, rejections."""
import ctypes as C
import struct

import numpy as np
import pytest

import ggml_whisper_util as G
import oracle_lib as O
import trackiellm_amd as tk
from trackiellm_amd.audio import WhisperHP


def probe(path):
    hp, nt, nn = WhisperHP(), C.c_int32(), C.c_int32()
    rc = tk.lib().tk_mi355x_whisper_ggml_probe(str(path).encode(), C.byref(hp), C.byref(nt), C.byref(nn))
    return rc, hp, nt.value, nn.value


@pytest.fixture(scope="module")
def small():
    ohp = O.whisper_tiny_test()
    orc = O.OracleWhisper(ohp, seed=6)
    T = orc.tensors()
    rounded, file_t = G.checkpoint_tensors(T, ohp)
    vocab = [bytes([65 + i % 26]) * (1 + i % 3) for i in range(ohp.n_vocab - 7)]
    return ohp, T, file_t, vocab


def test_probe_reads_geometry_vocabulary_and_directory(small, tmp_path):
    ohp, T, file_t, vocab = small
    p = tmp_path / "ggml-test.bin"
    G.write_ggml(p, ohp, T["frontend.mel_filters"], vocab, file_t)
    rc, hp, n_tok, n_tensors = probe(p)
    assert rc == 0, tk.lib().tk_error_get_detail()
    assert [getattr(hp, n) for n, _ in hp._fields_] == [getattr(ohp, n) for n, _ in ohp._fields_]
    assert n_tok == len(vocab) and n_tensors == len(file_t)
    # f32 checkpoints (ftype 0) parse too
    p32 = tmp_path / "ggml-test-f32.bin"
    G.write_ggml(p32, ohp, T["frontend.mel_filters"], vocab, [(n, a, 0) for n, a, _ in file_t], ftype=0)
    assert probe(p32)[0] == 0


def test_rejections(small, tmp_path):
    ohp, T, file_t, vocab = small
    mf = T["frontend.mel_filters"]
    bad = tmp_path / "bad.bin"
    G.write_ggml(bad, ohp, mf, vocab, file_t, magic=0x46554747)          # a GGUF magic, not ggml
    assert probe(bad)[0] != 0
    G.write_ggml(bad, ohp, mf, vocab, file_t[:-1])                        # decoder.ln.bias missing
    assert probe(bad)[0] != 0 and b"missing" in tk.lib().tk_error_get_detail()
    G.write_ggml(bad, ohp, mf, vocab, [(n, a, 2 if i == 3 else t) for i, (n, a, t) in enumerate(file_t)])  # a quantised tensor type
    assert probe(bad)[0] != 0 and b"quantised" in tk.lib().tk_error_get_detail()
    G.write_ggml(bad, ohp, mf[:, :100], vocab, file_t)                    # filter bank of the wrong width
    assert probe(bad)[0] != 0
    good = tmp_path / "good.bin"
    G.write_ggml(good, ohp, mf, vocab, file_t)
    data = open(good, "rb").read()
    open(bad, "wb").write(data[: len(data) - 10])                        # truncated last tensor
    assert probe(bad)[0] != 0
    open(bad, "wb").write(data[:30])                                      # truncated header
    assert probe(bad)[0] != 0
    wrong = [(n, (a[:-1] if n == "decoder.ln.weight" else a), t) for n, a, t in file_t]
    G.write_ggml(bad, ohp, mf, vocab, wrong)                              # wrong element count
    assert probe(bad)[0] != 0
    assert probe(tmp_path / "nope.bin")[0] != 0


def test_crafted_tensor_shape_cannot_wrap_the_bounds_check(small, tmp_path):
    """ne values whose product overflows int64 (or exceeds the file) must be rejected, not wrapped past `offset + bytes > fsize`"""
    import struct
    import trackiellm_amd as tk
    ohp, T, file_t, vocab = small
    p = tmp_path / "ok.bin"
    G.write_ggml(p, ohp, T["frontend.mel_filters"], vocab, file_t)
    raw = bytearray(p.read_bytes())
    name = file_t[0][0].encode()
    at = raw.index(name)               # tensor header = n_dims, name_len, type, ne[n_dims], then the name
    nd = file_t[0][1].ndim
    ne_at = at - 4 * nd
    for k in range(nd):
        struct.pack_into("<i", raw, ne_at + 4 * k, 0x7FFFFFFF)
    bad = tmp_path / "wrap.bin"
    bad.write_bytes(bytes(raw))
    hp = tk.WhisperHP()
    rc = tk.lib().tk_mi355x_whisper_ggml_probe(str(bad).encode(), C.byref(hp), None, None)
    assert rc == 4000  # TK_ERROR_MODEL_LOAD_FAILED
