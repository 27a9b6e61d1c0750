"""LoRA adapters (the reference applies one at load: src/ai_models/tk_model_loader.c:259-270): container readers and the oracle's merge. No GPU."""
import numpy as np
import pytest

import gguf_util
import oracle_lib as O


def factors(cfg, rng, r, which=((0, 1), (1, 3), (1, 8), (0, 6), (-1, O.T_OUTPUT))):
    D, QD, KVD, FF, V = cfg.d_model, cfg.n_head * cfg.head_dim, cfg.n_kv_head * cfg.head_dim, cfg.d_ff, cfg.vocab
    shape = {1: (QD, D), 2: (KVD, D), 3: (KVD, D), 4: (D, QD), 6: (FF, D), 7: (FF, D), 8: (D, FF)}
    out = {}
    for layer, w in which:
        n, k = (V, D) if layer < 0 else shape[w]
        out[(layer, w)] = (rng.normal(0, 0.05, (r, k)).astype(np.float32), rng.normal(0, 0.05, (n, r)).astype(np.float32))
    return out


def test_adapter_containers_are_read(tk, tmp_path):
    cfg = O.tiny_config()
    rng = np.random.default_rng(1)
    fs = factors(cfg, rng, 8)
    a, b = str(tmp_path / "a.ggla"), str(tmp_path / "b.gguf")
    gguf_util.write_lora_ggla(a, 8, 16, fs)
    gguf_util.write_lora_gguf(b, 16.0, fs, f16=True)
    assert tk.lora_probe(a) == (8, 16.0, 5)
    assert tk.lora_probe(b) == (8, 16.0, 5)
    raw = open(a, "rb").read()
    for name, data in (("cut", raw[:len(raw) - 100]), ("magic", b"ggml" + raw[4:]), ("version", raw[:4] + b"\x02\0\0\0" + raw[8:]), ("tiny", raw[:6])):
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        with pytest.raises(tk.TkError):
            tk.lora_probe(p)
    lone = dict(list(fs.items())[:1])
    p = str(tmp_path / "lone.gguf")
    gguf_util.write_lora_gguf(p, 16.0, lone)
    raw = open(p, "rb").read().replace(b".lora_b", b".lora_c")                  # factor A without factor B
    open(p, "wb").write(raw)
    with pytest.raises(tk.TkError):
        tk.lora_probe(p)
    p = str(tmp_path / "notadapter.gguf")
    gguf_util.write_gguf(p, [("general.architecture", gguf_util.GGUF_STRING, "llama")], [])
    with pytest.raises(tk.TkError):
        tk.lora_probe(p)
    with pytest.raises(tk.TkError):
        tk.lora_probe(str(tmp_path / "absent.bin"))


def test_oracle_merge_is_w_plus_scaled_ba():
    """dequant(merged) = dequant(base) + scale B A up to the block quantiser's step, for Q4_K, Q6_K and f16 tensors; a zero adapter moves nothing
    beyond a re-quantisation of the same values"""
    cfg = O.tiny_config()
    rng = np.random.default_rng(2)
    fs = factors(cfg, rng, 4)
    for f16 in (False, True):
        base = O.OracleLlm(cfg, seed=4, f16=f16)
        merged = O.OracleLlm(cfg, seed=4, f16=f16)
        for (layer, w), (A, B) in fs.items():
            n, k = B.shape[0], A.shape[1]
            w0 = base.dequant(layer, w, n, k).astype(np.float64)
            merged.apply_lora(layer, w, A, B, 2.0)
            w1 = merged.dequant(layer, w, n, k).astype(np.float64)
            want = w0 + 2.0 * (B.astype(np.float64) @ A.astype(np.float64))
            t, _ = merged.get_tensor(layer, w)
            tol = {O.TYPE_Q4_K: 0.08, O.TYPE_Q6_K: 0.02, O.TYPE_F16: 1e-3}[t]
            assert np.abs(w1 - want).max() < tol * np.abs(want).max(), (layer, w, t)
            assert np.abs(w1 - w0).max() > 10 * np.abs(w1 - want).max() or t == O.TYPE_Q4_K      # the adapter is what moved the tensor
        z = O.OracleLlm(cfg, seed=4, f16=f16)
        A, B = fs[(0, 1)]
        z.apply_lora(0, 1, A, np.zeros_like(B), 2.0)
        w0 = base.dequant(0, 1, B.shape[0], A.shape[1])
        wz = z.dequant(0, 1, B.shape[0], A.shape[1])
        assert np.abs(wz - w0).max() <= (0.0 if f16 else 0.08 * np.abs(w0).max())
