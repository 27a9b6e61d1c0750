"""Minimal ONNX writer for tests: a ModelProto whose graph holds Conv nodes + their float initialisers, hand-encoded in the protobuf
wire format (field numbers of onnx.proto3).  Mirrors what an Ultralytics YOLOv8 export looks like to a reader that only wants the
convolution weights: Conv nodes in execution order, weights [cout][cin][kh][kw], then the constant DFL 1x1 conv."""
import struct

import numpy as np


def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _vi(field, v):
    return _varint(field << 3) + _varint(v)


def tensor(name, arr, raw=True, f16=False):
    arr = np.ascontiguousarray(arr, np.float16 if f16 else np.float32)
    t = b"".join(_vi(1, int(d)) for d in arr.shape) + _vi(2, 10 if f16 else 1)
    if raw or f16:
        t += _ld(9, arr.tobytes())
    else:
        t += _ld(4, arr.astype("<f4").tobytes())  # packed float_data
    return t + _ld(8, name.encode())


def conv_node(x, w, b, y):
    n = _ld(1, x.encode()) + _ld(1, w.encode())
    if b:
        n += _ld(1, b.encode())
    return n + _ld(2, y.encode()) + _ld(4, b"Conv")


def other_node(op, x, y):
    return _ld(1, x.encode()) + _ld(2, y.encode()) + _ld(4, op.encode())


def yolo_model(layers, with_dfl=True, raw=True, f16=False, drop_bias_of=None):
    """layers: oracle_lib.OracleYolo.layers() (w [cout][k][k][cin], b [cout]) -> bytes of an .onnx file"""
    g = b""
    inits = b""
    prev = "images"
    for i, L in enumerate(layers):
        w = np.ascontiguousarray(L["w"].transpose(0, 3, 1, 2))  # -> [cout][cin][kh][kw]
        wn, bn = "model.%d.conv.weight" % i, "model.%d.conv.bias" % i
        inits += _ld(5, tensor(wn, w, raw=(raw if i % 2 else True), f16=f16))
        has_b = drop_bias_of != i
        if has_b:
            inits += _ld(5, tensor(bn, L["b"], raw=raw))
        g += _ld(1, conv_node(prev, wn, bn if has_b else "", "c%d" % i))
        g += _ld(1, other_node("Sigmoid", "c%d" % i, "s%d" % i))  # a non-Conv node in between, to be ignored
        prev = "s%d" % i
    if with_dfl:
        dfl = np.arange(16, dtype=np.float32).reshape(1, 16, 1, 1)
        inits += _ld(5, tensor("model.22.dfl.conv.weight", dfl))
        g += _ld(1, conv_node(prev, "model.22.dfl.conv.weight", "", "dfl"))
    # an unrelated int64 initialiser (shape constants of Reshape nodes look like this) must be skipped silently
    shape_t = _vi(1, 2) + _vi(2, 7) + _ld(9, struct.pack("<2q", 1, -1)) + _ld(8, b"shape_const")
    inits += _ld(5, shape_t)
    graph = g + _ld(2, b"torch_jit") + inits
    return _vi(1, 8) + _ld(2, b"pytorch") + _ld(7, graph) + _ld(8, _ld(1, b"") + _vi(2, 17))
