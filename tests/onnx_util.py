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


def attr_float(name, v):
    return _ld(1, name.encode()) + _varint((2 << 3) | 5) + struct.pack("<f", v) + _vi(20, 1)


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


# ---- a depth network of the convolutional MiDaS class (EfficientNet-lite-like encoder, feature-fusion decoder), small enough for fixtures ----

def _depth_dims(width):
    """channel counts of the depth graph; width = 1 is the test fixture's size, width = 4 is close to MiDaS v2.1 small in work per pixel"""
    w = width
    return {"stem": 16 * w, "b1": 24 * w, "f4": 32 * w, "dec": 24 * w, "d2": 16 * w, "h1": 8 * w}


def depth_weights(seed, width=1):
    rng = np.random.default_rng(seed)
    D = _depth_dims(width)
    def conv(m, c, k, gain=1.0):
        return (rng.standard_normal((m, c, k, k)) * gain / np.sqrt(c * k * k)).astype(np.float32), (rng.standard_normal(m) * 0.1).astype(np.float32)
    W = {}
    shapes = {"stem": (D["stem"], 3, 3), "dw1": (D["stem"], 1, 3), "pw1": (D["b1"], D["stem"], 1), "down": (D["f4"], D["b1"] + 2 * D["stem"], 3),
              "dw2": (D["f4"], 1, 3), "pw2": (D["f4"], D["f4"], 1), "red": (D["dec"], D["f4"], 1), "lat": (D["dec"], D["b1"], 1),
              "fuse": (D["d2"], D["dec"], 3), "head1": (D["h1"], D["d2"], 3), "head2": (1, D["h1"], 1)}
    for name, (m, c, k) in shapes.items():
        W[name + ".w"], W[name + ".b"] = conv(m, c, k, 1.6)
    n = D["stem"]
    W["bn.scale"] = (1.0 + 0.2 * rng.standard_normal(n)).astype(np.float32)
    W["bn.bias"] = (0.1 * rng.standard_normal(n)).astype(np.float32)
    W["bn.mean"] = (0.2 * rng.standard_normal(n)).astype(np.float32)
    W["bn.var"] = (0.5 + rng.random(n)).astype(np.float32)
    W["head2.b"] = np.array([0.3], np.float32)
    W["c0"] = np.array(0.0, np.float32)
    W["c6"] = np.array(6.0, np.float32)
    W["up_scales"] = np.array([1, 1, 2, 2], np.float32)
    W["roi"] = np.zeros(0, np.float32)
    return W


def depth_spec(width=1):
    """nodes of the network as dicts (the oracle and the torch fixture script evaluate this list; depth_model() writes it as ONNX)"""
    D = _depth_dims(width)
    def n(op, i, o, **attrs):
        return {"op": op, "in": i, "out": [o], "attrs": attrs}
    p1 = [1, 1, 1, 1]
    return [
        n("Conv", ["input", "stem.w", "stem.b"], "s0", strides=[2, 2], pads=p1),
        n("BatchNormalization", ["s0", "bn.scale", "bn.bias", "bn.mean", "bn.var"], "s1", epsilon=1e-3),
        n("Clip", ["s1", "c0", "c6"], "f2"),
        n("Conv", ["f2", "dw1.w", "dw1.b"], "d1", pads=p1, group=D["stem"]),
        n("Clip", ["d1"], "d1c", min=0.0, max=6.0),
        n("Conv", ["d1c", "pw1.w", "pw1.b"], "b1"),
        n("MaxPool", ["f2"], "mp", kernel_shape=[3, 3], strides=[1, 1], pads=p1),
        n("AveragePool", ["f2"], "ap", kernel_shape=[3, 3], strides=[1, 1], pads=p1, count_include_pad=0),
        n("Concat", ["b1", "mp", "ap"], "cat", axis=1),
        n("Conv", ["cat", "down.w", "down.b"], "f4p", strides=[2, 2], pads=p1),
        n("Relu", ["f4p"], "f4"),
        n("Conv", ["f4", "dw2.w", "dw2.b"], "d2", pads=p1, group=D["f4"]),
        n("LeakyRelu", ["d2"], "d2a", alpha=0.1),
        n("Conv", ["d2a", "pw2.w", "pw2.b"], "p2"),
        n("Add", ["p2", "f4"], "b2r"),
        n("GlobalAveragePool", ["b2r"], "se"),
        n("Sigmoid", ["se"], "seg"),
        n("Mul", ["b2r", "seg"], "b2"),
        n("Conv", ["b2", "red.w", "red.b"], "r4"),
        n("Resize", ["r4", "roi", "up_scales"], "up1", mode="linear", coordinate_transformation_mode="align_corners"),
        n("Conv", ["b1", "lat.w", "lat.b"], "l1"),
        n("Add", ["up1", "l1"], "m2"),
        n("Conv", ["m2", "fuse.w", "fuse.b"], "d2p", pads=p1),
        n("Relu", ["d2p"], "dd2"),
        n("Resize", ["dd2", "roi", "up_scales"], "up2", mode="linear", coordinate_transformation_mode="half_pixel"),
        n("Pad", ["up2", "pad1"], "up2p", mode="constant"),
        n("Conv", ["up2p", "head1.w", "head1.b"], "h1"),
        n("Relu", ["h1"], "h1r"),
        n("Conv", ["h1r", "head2.w", "head2.b"], "h2"),
        n("Relu", ["h2"], "h2r"),
        n("Squeeze", ["h2r"], "output", axes=[1]),
    ]


def depth_consts(W):
    c = dict(W)
    c["pad1"] = np.array([0, 0, 1, 1, 0, 0, 1, 1], np.int64)
    return c


def depth_model(W, height=-1, width=-1, extra_op=None, channels=1):
    nodes = []
    for nd in depth_spec(channels):
        attrs = []
        for k, v in nd["attrs"].items():
            if isinstance(v, str):
                attrs.append(attr_str(k, v))
            elif isinstance(v, float):
                attrs.append(attr_float(k, v))
            elif isinstance(v, int):
                attrs.append(attr_int(k, v))
            else:
                attrs.append(attr_ints(k, v))
        nodes.append(node(nd["op"], nd["in"], nd["out"], attrs, name=nd["out"][0] + "_node"))
    if extra_op:
        nodes.append(node(extra_op, ["output"], ["unused"], name="extra"))
    inits = [tensor(k, v) for k, v in W.items()] + [int_tensor("pad1", [0, 0, 1, 1, 0, 0, 1, 1])]
    return model(nodes, inits, [value_info("input", 1, [1, 3, height, width])], [value_info("output", 1, [1, height, width])])
