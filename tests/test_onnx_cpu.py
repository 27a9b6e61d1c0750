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


def test_vad_graph_probe_accepts_the_silero_class_and_names_unsupported_ops(tmp_path):
    """general ONNX graph reader (csrc/nn/tk_onnx_graph.cpp): nodes, attributes, float / int64 initialisers, declared inputs; the VAD
    executor's op check runs without a GPU"""
    W = X.vad_weights(21)
    ok = tmp_path / "vad.onnx"
    ok.write_bytes(X.vad_model(W))
    n, ni, ns = C.c_int32(), C.c_int32(), C.c_int32()
    assert tk.lib().tk_mi355x_vad_onnx_probe(str(ok).encode(), C.byref(n), C.byref(ni), C.byref(ns)) == 0
    assert n.value == 24 and ni.value == len(W) + 5 and ns.value == 2          # h and c are recurrent inputs
    bad = tmp_path / "vad_einsum.onnx"
    bad.write_bytes(X.vad_model(W, extra_op="Einsum"))
    assert tk.lib().tk_mi355x_vad_onnx_probe(str(bad).encode(), None, None, None) == 4001   # TK_ERROR_MODEL_VERIFICATION_FAILED
    assert b"'Einsum'" in tk.lib().tk_error_get_detail()
    nobody = tmp_path / "vad_loop.onnx"
    nobody.write_bytes(X.vad_model(W, extra_op="Loop"))                          # a Loop without its body graph
    assert tk.lib().tk_mi355x_vad_onnx_probe(str(nobody).encode(), None, None, None) == 4001 and b"body" in tk.lib().tk_error_get_detail()
    # If with its two branch graphs (what per-sample-rate exports use) is read, its branches' ops are checked too
    sw = tmp_path / "vad_if.onnx"
    sw.write_bytes(X.vad_model(W, with_if=True))
    assert tk.lib().tk_mi355x_vad_onnx_probe(str(sw).encode(), C.byref(n), C.byref(ni), C.byref(ns)) == 0
    assert n.value == 24 and ns.value == 2                                      # Equal + If replace the two head nodes of the plain graph
    naked = tmp_path / "vad_if_nobranch.onnx"
    naked.write_bytes(X.vad_model(W, extra_op="If"))                             # an If without branch graphs
    assert tk.lib().tk_mi355x_vad_onnx_probe(str(naked).encode(), None, None, None) == 4001 and b"then_branch" in tk.lib().tk_error_get_detail()
    assert tk.lib().tk_mi355x_vad_onnx_probe(str(tmp_path / "none.onnx").encode(), None, None, None) == 3001
    junk = tmp_path / "junk.onnx"
    junk.write_bytes(b"\x3a\xff\xff\xff\xff\x0f" + b"\0" * 16)                   # a graph field longer than the file
    assert tk.lib().tk_mi355x_vad_onnx_probe(str(junk).encode(), None, None, None) == 3004


def test_loop_and_scan_graphs_pass_the_op_check_and_malformed_ones_do_not(tmp_path):
    """Loop / Scan bodies are read as sub-graphs, their nodes checked like the outer ones, their arity against the node's (no GPU)"""
    W = X.loopnet_weights(3)
    n = C.c_int32()
    for mode in ("count", "cond"):
        p = tmp_path / ("loop_%s.onnx" % mode)
        p.write_bytes(X.loopnet_model(W, X.loopnet_spec(mode)))
        assert tk.lib().tk_mi355x_depth_onnx_probe(str(p).encode(), C.byref(n)) == 0, tk.lib().tk_error_get_detail()
    bad = X.loopnet_spec("count")
    bad[1]["in"] = ["trip", "bool_go", "y_0", "y_0"]
    p = tmp_path / "arity.onnx"
    p.write_bytes(X.loopnet_model(W, bad))
    assert tk.lib().tk_mi355x_depth_onnx_probe(str(p).encode(), C.byref(n)) != 0 and b"body" in tk.lib().tk_error_get_detail()
    bad = X.loopnet_spec("count")
    bad[3]["attrs"].pop("num_scan_inputs")
    p.write_bytes(X.loopnet_model(W, bad))
    assert tk.lib().tk_mi355x_depth_onnx_probe(str(p).encode(), C.byref(n)) != 0
