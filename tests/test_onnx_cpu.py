"""ONNX Conv-initialiser extraction for the detector, parsing only (no GPU)."""
import ctypes as C

import numpy as np
import pytest

import onnx_util as X
import oracle_lib as O
import trackiellm_amd as tk


def probe(path):
    n, p = C.c_int32(), C.c_int64()
    rc = tk.lib().tk_mi355x_onnx_probe(str(path).encode(), C.byref(n), C.byref(p))
    return rc, n.value, p.value


@pytest.fixture(scope="module")
def layers():
    return O.OracleYolo(nc=80, seed=5, cls_bias=-4.0).layers()


def test_probe_counts_convs_and_parameters(layers, tmp_path):
    want_params = sum(L["w"].size + L["b"].size for L in layers)
    for kw in (dict(), dict(raw=False), dict(f16=True), dict(with_dfl=False)):
        p = tmp_path / "y.onnx"
        p.write_bytes(X.yolo_model(layers, **kw))
        rc, n, params = probe(p)
        assert rc == 0, (kw, tk.lib().tk_error_get_detail())
        dfl = 0 if kw.get("with_dfl") is False else 1
        assert n == 63 + dfl and params == want_params + 17 * dfl  # DFL: 16 weights + the zero bias the reader adds


def test_rejections(layers, tmp_path):
    p = tmp_path / "bad.onnx"
    p.write_bytes(X.yolo_model(layers[:-1]))                      # one conv short
    assert probe(p)[0] != 0
    swapped = list(layers)
    swapped[0], swapped[1] = swapped[1], swapped[0]
    p.write_bytes(X.yolo_model(swapped))                          # shapes out of graph order
    assert probe(p)[0] != 0 and b"does not match" in tk.lib().tk_error_get_detail()
    good = X.yolo_model(layers)
    p.write_bytes(good[: len(good) // 2])                         # truncated
    assert probe(p)[0] != 0
    p.write_bytes(b"GGUF" + bytes(100))                           # not protobuf / no graph
    assert probe(p)[0] != 0
    assert probe(tmp_path / "missing.onnx")[0] != 0
    # a Conv without a bias input is legal ONNX: the bias reads as zeros
    p.write_bytes(X.yolo_model(layers, drop_bias_of=5))
    assert probe(p)[0] == 0
