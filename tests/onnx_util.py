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


# ---- general graphs (nodes with attributes, int64 initialisers, declared inputs / outputs): used for the VAD graph tests ----

def int_tensor(name, values, dims=None):
    values = np.asarray(values, np.int64).reshape(-1)
    dims = [len(values)] if dims is None else dims
    t = b"".join(_vi(1, int(d)) for d in dims) + _vi(2, 7) + _ld(9, values.astype("<i8").tobytes())
    return t + _ld(8, name.encode())


def attr_int(name, v):
    return _ld(1, name.encode()) + _vi(3, v & 0xFFFFFFFFFFFFFFFF) + _vi(20, 2)


def attr_ints(name, vals):
    return _ld(1, name.encode()) + b"".join(_vi(8, v & 0xFFFFFFFFFFFFFFFF) for v in vals) + _vi(20, 7)


def attr_str(name, s):
    return _ld(1, name.encode()) + _ld(4, s.encode()) + _vi(20, 3)


def node(op, inputs, outputs, attrs=(), name=""):
    n = b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs)
    if name:
        n += _ld(3, name.encode())
    return n + _ld(4, op.encode()) + b"".join(_ld(5, a) for a in attrs)


def value_info(name, elem_type, dims):
    shape = b"".join(_ld(1, _vi(1, d) if d >= 0 else _ld(2, b"N")) for d in dims)
    tensor_type = _vi(1, elem_type) + _ld(2, shape)
    return _ld(1, name.encode()) + _ld(2, _ld(1, tensor_type))


def model(nodes, initialisers, inputs, outputs):
    graph = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"g") + b"".join(_ld(5, t) for t in initialisers)
    graph += b"".join(_ld(11, i) for i in inputs) + b"".join(_ld(12, o) for o in outputs)
    return _vi(1, 8) + _ld(2, b"test") + _ld(7, graph) + _ld(8, _ld(1, b"") + _vi(2, 17))


def vad_weights(seed, n_bins=33, n_fft=64, hidden=32):
    """seeded parameters of the generated Silero-class VAD graph (see vad_model)"""
    rng = np.random.default_rng(seed)
    k = np.arange(n_fft)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * k / n_fft)
    basis = np.concatenate([np.cos(2 * np.pi * np.outer(np.arange(n_bins), k) / n_fft) * win, -np.sin(2 * np.pi * np.outer(np.arange(n_bins), k) / n_fft) * win])
    W = {"stft": basis[:, None, :].astype(np.float32)}
    for name, (co, ci, kk) in {"enc1": (32, n_bins, 3), "enc2": (32, 32, 3), "enc3": (hidden, 32, 3)}.items():
        W[name + ".w"] = (rng.standard_normal((co, ci, kk)) * np.sqrt(2.0 / (ci * kk))).astype(np.float32)
        W[name + ".b"] = (0.1 * rng.standard_normal(co)).astype(np.float32)
    W["lstm.W"] = (rng.standard_normal((1, 4 * hidden, hidden)) * 0.3).astype(np.float32)
    W["lstm.R"] = (rng.standard_normal((1, 4 * hidden, hidden)) * 0.3).astype(np.float32)
    W["lstm.B"] = (0.1 * rng.standard_normal((1, 8 * hidden))).astype(np.float32)
    W["head.w"] = (rng.standard_normal((1, hidden, 1)) * 0.5).astype(np.float32)
    W["head.b"] = np.array([-0.2], np.float32)
    return W


def vad_model(W, window=480, hop=32, hidden=32, extra_op=None):
    """A VAD graph of the Silero class: reflect pad -> STFT as a strided Conv with a fixed (windowed cos | -sin) basis -> magnitude ->
    three Conv + ReLU (two strided) -> LSTM with recurrent inputs h, c -> ReLU -> 1x1 Conv -> Sigmoid -> mean over time.
    Inputs: input [1, window] f32, sr [] i64, h [1, 1, hidden], c [1, 1, hidden]; outputs: output [1, 1], hn, cn."""
    n_bins = W["stft"].shape[0] // 2
    n_fft = W["stft"].shape[2]
    nodes = [
        node("Unsqueeze", ["input"], ["x3"], [attr_ints("axes", [1])]),
        node("Pad", ["x3", "pads"], ["xp"], [attr_str("mode", "reflect")]),
        node("Conv", ["xp", "stft"], ["spec"], [attr_ints("strides", [hop]), attr_ints("kernel_shape", [n_fft])]),
        node("Slice", ["spec", "s0", "s1", "ax1"], ["re"]),
        node("Slice", ["spec", "s1", "s2", "ax1"], ["im"]),
        node("Constant", [], ["two"], [_ld(1, b"value") + _ld(5, tensor("", np.array([2.0], np.float32))) + _vi(20, 4)]),
        node("Pow", ["re", "two"], ["re2"]),
        node("Pow", ["im", "two"], ["im2"]),
        node("Add", ["re2", "im2"], ["pw"]),
        node("Sqrt", ["pw"], ["mag"]),
        node("Conv", ["mag", "enc1.w", "enc1.b"], ["e1"], [attr_ints("pads", [1, 1]), attr_ints("strides", [1])]),
        node("Relu", ["e1"], ["r1"]),
        node("Conv", ["r1", "enc2.w", "enc2.b"], ["e2"], [attr_ints("pads", [1, 1]), attr_ints("strides", [2])]),
        node("Relu", ["e2"], ["r2"]),
        node("Conv", ["r2", "enc3.w", "enc3.b"], ["e3"], [attr_ints("pads", [1, 1]), attr_ints("strides", [2])]),
        node("Relu", ["e3"], ["r3"]),
        node("Transpose", ["r3"], ["seq"], [attr_ints("perm", [2, 0, 1])]),
        node("LSTM", ["seq", "lstm.W", "lstm.R", "lstm.B", "", "h", "c"], ["y", "hn", "cn"], [attr_int("hidden_size", hidden)], name="lstm"),
        node("Squeeze", ["y"], ["y3"], [attr_ints("axes", [1])]),
        node("Transpose", ["y3"], ["yt"], [attr_ints("perm", [1, 2, 0])]),
        node("Relu", ["yt"], ["yr"]),
        node("Conv", ["yr", "head.w", "head.b"], ["logit"]),
        node("Sigmoid", ["logit"], ["p"]),
        node("ReduceMean", ["p"], ["output"], [attr_ints("axes", [2]), attr_int("keepdims", 0)]),
    ]
    if extra_op:
        nodes.append(node(extra_op, ["output"], ["unused"], name="extra"))
    inits = [tensor(k, v) for k, v in W.items()]
    inits += [int_tensor("pads", [0, 0, n_fft // 2, 0, 0, n_fft // 2]), int_tensor("s0", [0]), int_tensor("s1", [n_bins]), int_tensor("s2", [2 * n_bins]),
              int_tensor("ax1", [1])]
    inputs = [value_info("input", 1, [1, window]), value_info("sr", 7, []), value_info("h", 1, [1, 1, hidden]), value_info("c", 1, [1, 1, hidden])]
    outputs = [value_info("output", 1, [1, 1]), value_info("hn", 1, [1, 1, hidden]), value_info("cn", 1, [1, 1, hidden])]
    return model(nodes, inits, inputs, outputs)
