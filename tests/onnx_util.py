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


def bool_tensor(name, values, dims=None):
    values = np.asarray(values).reshape(-1).astype(np.uint8)
    dims = [len(values)] if dims is None else dims
    return b"".join(_vi(1, int(d)) for d in dims) + _vi(2, 9) + _ld(9, values.tobytes()) + _ld(8, name.encode())


def attr_int(name, v):
    return _ld(1, name.encode()) + _vi(3, v & 0xFFFFFFFFFFFFFFFF) + _vi(20, 2)


def attr_ints(name, vals):
    return _ld(1, name.encode()) + b"".join(_vi(8, v & 0xFFFFFFFFFFFFFFFF) for v in vals) + _vi(20, 7)


def attr_float(name, v):
    return _ld(1, name.encode()) + _varint((2 << 3) | 5) + struct.pack("<f", v) + _vi(20, 1)


def attr_graph(name, graph_bytes):
    """a sub-graph attribute (AttributeProto.g = 6, type GRAPH = 5): the then_branch / else_branch of If"""
    return _ld(1, name.encode()) + _ld(6, graph_bytes) + _vi(20, 5)


def graph_proto(nodes, initialisers, inputs, outputs, name=b"g"):
    g = b"".join(_ld(1, n) for n in nodes) + _ld(2, name) + b"".join(_ld(5, t) for t in initialisers)
    return g + b"".join(_ld(11, i) for i in inputs) + b"".join(_ld(12, o) for o in outputs)


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


def vad_model(W, window=480, hop=32, hidden=32, extra_op=None, with_if=False, neg_head=False):
    """with_if: the head sits inside an If on Equal(sr, 16000) — the way published silero_vad.onnx exports select their per-sample-rate
    sub-network (to my knowledge of those files; none is available offline): then_branch = the head below, else_branch = the same head on
    the negated features (an initialiser of its own inside the branch).  neg_head: the else-branch's arithmetic as a plain graph.
    A VAD graph of the Silero class: reflect pad -> STFT as a strided Conv with a fixed (windowed cos | -sin) basis -> magnitude ->
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
    ]
    neg = [node("Mul", ["yr", "minus_one"], ["yn"]), node("Conv", ["yn", "head.w", "head.b"], ["logit_e"]), node("Sigmoid", ["logit_e"], ["p_e"])]
    if with_if:
        then_g = graph_proto([node("Conv", ["yr", "head.w", "head.b"], ["logit_t"]), node("Sigmoid", ["logit_t"], ["p_t"])], [], [],
                             [value_info("p_t", 1, [1, 1, -1])], b"then")
        else_g = graph_proto(neg, [tensor("minus_one", np.array([-1.0], np.float32))], [], [value_info("p_e", 1, [1, 1, -1])], b"else")
        nodes += [node("Equal", ["sr", "sr16k"], ["is16k"]),
                  node("If", ["is16k"], ["p"], [attr_graph("then_branch", then_g), attr_graph("else_branch", else_g)], name="rate_switch")]
    elif neg_head:
        nodes += neg[:2] + [node("Sigmoid", ["logit_e"], ["p"])]
    else:
        nodes += [node("Conv", ["yr", "head.w", "head.b"], ["logit"]), node("Sigmoid", ["logit"], ["p"])]
    nodes.append(node("ReduceMean", ["p"], ["output"], [attr_ints("axes", [2]), attr_int("keepdims", 0)]))
    if extra_op:
        nodes.append(node(extra_op, ["output"], ["unused"], name="extra"))
    inits = [tensor(k, v) for k, v in W.items()]
    inits += [int_tensor("pads", [0, 0, n_fft // 2, 0, 0, n_fft // 2]), int_tensor("s0", [0]), int_tensor("s1", [n_bins]), int_tensor("s2", [2 * n_bins]),
              int_tensor("ax1", [1])]
    if with_if:
        inits.append(int_tensor("sr16k", [16000], []))
    if neg_head:
        inits.append(tensor("minus_one", np.array([-1.0], np.float32)))
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


# ---- a depth network of the DPT / Swin-transformer class (the model the reference names: DPT-SwinV2-Tiny-256, src/vision/tk_depth_midas.c:8,
# tests/tk_cortex_test.cpp:42), small enough for fixtures.  Structure after the published SwinV2 + DPT definitions as an ONNX export spells
# them: patch embedding (strided Conv -> tokens -> LayerNormalization), window attention with cosine similarity (ReduceL2 / Clip / Max, Div),
# a learned logit scale, a relative-position bias gathered from a table (Gather -> Reshape -> Transpose -> Sigmoid * 16), shifted windows
# (roll as Slice + Concat, the mask through Where), post-norm residuals, an MLP with GELU spelled with Erf (and once as the Gelu op), patch
# merging (four strided Slices -> Concat -> MatMul -> LayerNormalization), a squeeze gate (ReduceSum), and a DPT-style head: tokens back to
# feature maps (Transpose + Reshape, one of them with a Shape-driven dynamic target), 1x1 / 3x3 Conv, Resize (align_corners), ConvTranspose
# (overlapping k3 s2 and non-overlapping k2 s2), Relu.  Not a real checkpoint's graph: there is none offline. ----

SWIN = dict(H=64, patch=4, C=16, heads=2, ws=4, shift=2, eps=1e-5)


def _rel_index(ws):
    """relative position index of a ws x ws window, as Swin computes it: [N, N] entries in [0, (2 ws - 1)^2)"""
    coords = np.stack(np.meshgrid(np.arange(ws), np.arange(ws), indexing="ij")).reshape(2, -1)
    rel = coords[:, :, None] - coords[:, None, :] + (ws - 1)
    return (rel[0] * (2 * ws - 1) + rel[1]).astype(np.int64)


def _shift_mask(hw, ws, shift):
    """Swin's attention mask of a shifted-window block: [nW, N, N], 1 where two tokens of a window come from different image regions"""
    img = np.zeros((hw, hw), np.int64)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[hs, wsl] = cnt
            cnt += 1
    win = img.reshape(hw // ws, ws, hw // ws, ws).transpose(0, 2, 1, 3).reshape(-1, ws * ws)
    return (win[:, :, None] != win[:, None, :]).astype(np.int64)


def swin_weights(seed):
    rng = np.random.default_rng(seed)
    S = SWIN
    C, heads, ws = S["C"], S["heads"], S["ws"]
    W = {}

    def lin(name, k, n, gain=1.0):
        W[name + ".w"] = (rng.standard_normal((k, n)) * gain / np.sqrt(k)).astype(np.float32)
        W[name + ".b"] = (rng.standard_normal(n) * 0.1).astype(np.float32)

    def ln(name, n):
        W[name + ".g"] = (1.0 + 0.2 * rng.standard_normal(n)).astype(np.float32)
        W[name + ".b"] = (0.1 * rng.standard_normal(n)).astype(np.float32)

    def conv(name, m, c, k, gain=1.4):
        W[name + ".w"] = (rng.standard_normal((m, c, k, k)) * gain / np.sqrt(c * k * k)).astype(np.float32)
        W[name + ".b"] = (rng.standard_normal(m) * 0.1).astype(np.float32)

    conv("pe", C, 3, S["patch"])
    ln("ln0", C)
    for blk, c in (("b1", C), ("b2", C), ("b3", 2 * C)):
        lin(blk + ".qkv", c, 3 * c, 1.5)
        lin(blk + ".proj", c, c)
        ln(blk + ".ln1", c)
        lin(blk + ".fc1", c, 4 * c, 1.3)
        lin(blk + ".fc2", 4 * c, c)
        ln(blk + ".ln2", c)
        W[blk + ".logit_scale"] = (4.0 + rng.random((heads, 1, 1)) * 4.0).astype(np.float32)
        W[blk + ".rpb"] = rng.standard_normal(((2 * ws - 1) ** 2, heads)).astype(np.float32)
    W["red.w"] = (rng.standard_normal((4 * C, 2 * C)) / np.sqrt(4 * C)).astype(np.float32)
    ln("lnm", 2 * C)
    conv("r1", 8, C, 1)
    conv("r2", 8, 2 * C, 1)
    conv("fu", 8, 8, 3)
    W["up1.w"] = (rng.standard_normal((8, 8, 3, 3)) / np.sqrt(8 * 9 / 4)).astype(np.float32)   # ConvTranspose: [Cin][Cout][kh][kw]
    W["up1.b"] = (rng.standard_normal(8) * 0.1).astype(np.float32)
    W["up2.w"] = (rng.standard_normal((8, 8, 2, 2)) / np.sqrt(8)).astype(np.float32)
    conv("h1", 4, 8, 3)
    conv("h2", 1, 4, 1)
    W["h2.b"] = np.array([0.4], np.float32)
    W["sqrt2"] = np.array(np.sqrt(2.0), np.float32)
    W["one"] = np.array(1.0, np.float32)
    W["half"] = np.array(0.5, np.float32)
    W["sixteen"] = np.array(16.0, np.float32)
    W["eps12"] = np.array(1e-12, np.float32)
    W["neg100"] = np.array(-100.0, np.float32)
    W["inv_tokens"] = np.array(1.0 / 64.0, np.float32)
    W["up_scales"] = np.array([1, 1, 2, 2], np.float32)
    W["roi"] = np.zeros(0, np.float32)
    return W


def swin_ints():
    """the integer initialisers of the graph: Reshape targets, Slice bounds, Gather indices, the shifted-window mask"""
    S = SWIN
    C, heads, ws, shift = S["C"], S["heads"], S["ws"], S["shift"]
    hw = S["H"] // S["patch"]
    I = {"i0": (np.array(0), []), "i1": (np.array(1), []), "i2": (np.array(2), []), "ax0": ([0], None), "ax1": ([1], None), "ax2": ([2], None),
         "rel_index": (_rel_index(ws).reshape(-1), None), "mask": (_shift_mask(hw, ws, shift).reshape(-1), [1, (hw // ws) ** 2, 1, ws * ws, ws * ws]),
         "s_shift": ([shift], None), "s_zero": ([0], None), "s_end": ([1 << 30], None), "s_back": ([-shift], None),
         "pm_ax": ([1, 2], None), "pm_end": ([1 << 30, 1 << 30], None), "pm_step": ([2, 2], None),
         "pm00": ([0, 0], None), "pm10": ([1, 0], None), "pm01": ([0, 1], None), "pm11": ([1, 1], None), "tail_hw_c": ([hw, hw, C], None)}
    for tag, h, c in (("1", hw, C), ("2", hw // 2, 2 * C)):
        nw, n, hd = (h // ws) ** 2, ws * ws, c // heads
        I.update({"sh_img" + tag: ([1, h, h, c], None), "sh_part" + tag: ([1, h // ws, ws, h // ws, ws, c], None), "sh_win" + tag: ([nw, n, c], None),
                  "sh_qkv" + tag: ([nw, n, 3, heads, hd], None), "sh_rpb": ([n, n, heads], None), "sh_att5_" + tag: ([1, nw, heads, n, n], None),
                  "sh_att4_" + tag: ([nw, heads, n, n], None), "sh_merge" + tag: ([1, h // ws, h // ws, ws, ws, c], None), "sh_tok" + tag: ([1, h * h, c], None),
                  "sh_map" + tag: ([1, c, h, h], None)})
    I["sh_pm"] = ([1, (hw // 2) ** 2, 4 * C], None)
    I["sh_pe"] = ([1, C, hw * hw], None)
    return I


def swin_spec():
    """nodes as dicts (the numpy oracle and the torch fixture script evaluate this list; swin_model() writes it as ONNX)"""
    S = SWIN
    hw = S["H"] // S["patch"]
    out = []

    def n(op, i, o, **attrs):
        out.append({"op": op, "in": i, "out": [o], "attrs": attrs})
        return o

    def layer_norm(x, name, o):
        return n("LayerNormalization", [x, name + ".g", name + ".b"], o, axis=-1, epsilon=S["eps"])

    def roll(x, p, first, second, axis_name, axis):
        """torch.roll along one axis as an export spells it: two Slices and a Concat"""
        a = n("Slice", [x, first, "s_end", axis_name], p + "_a")
        b = n("Slice", [x, "s_zero", first, axis_name], p + "_b")
        return n("Concat", [a, b], p + "_r", axis=axis)

    def block(x, p, tag, shifted, gelu_op):
        xs = n("Reshape", [x, "sh_img" + tag], p + "_img")
        if shifted:  # roll by -shift on both image axes
            xs = roll(roll(xs, p + "_rh", "s_shift", None, "ax1", 1), p + "_rw", "s_shift", None, "ax2", 2)
        t = n("Reshape", [xs, "sh_part" + tag], p + "_part")
        t = n("Transpose", [t], p + "_partt", perm=[0, 1, 3, 2, 4, 5])
        win = n("Reshape", [t, "sh_win" + tag], p + "_win")
        qkv = n("Add", [n("MatMul", [win, p + ".qkv.w"], p + "_qkvm"), p + ".qkv.b"], p + "_qkv")
        t = n("Transpose", [n("Reshape", [qkv, "sh_qkv" + tag], p + "_qkvr")], p + "_qkvt", perm=[2, 0, 3, 1, 4])
        q, k, v = (n("Gather", [t, ix], p + "_" + nm, axis=0) for nm, ix in (("q", "i0"), ("k", "i1"), ("v", "i2")))
        qn = n("Div", [q, n("Clip", [n("ReduceL2", [q], p + "_qn2", axes=[-1], keepdims=1), "eps12"], p + "_qnc")], p + "_qn")
        kn = n("Div", [k, n("Max", [n("ReduceL2", [k], p + "_kn2", axes=[-1], keepdims=1), "eps12"], p + "_knc")], p + "_kn")
        att = n("MatMul", [qn, n("Transpose", [kn], p + "_kt", perm=[0, 1, 3, 2])], p + "_qk")
        att = n("Mul", [att, p + ".logit_scale"], p + "_qks")
        bias = n("Gather", [p + ".rpb", "rel_index"], p + "_rpbg", axis=0)
        bias = n("Transpose", [n("Reshape", [bias, "sh_rpb"], p + "_rpbr")], p + "_rpbt", perm=[2, 0, 1])
        bias = n("Unsqueeze", [n("Mul", [n("Sigmoid", [bias], p + "_rpbs"), "sixteen"], p + "_rpb16"), "ax0"], p + "_bias")
        att = n("Add", [att, bias], p + "_qkb")
        if shifted:  # masked_fill(mask, -100) through Where, on the [1, nW, heads, N, N] view
            a5 = n("Reshape", [att, "sh_att5_" + tag], p + "_a5")
            a5 = n("Where", ["mask", n("Add", [a5, "neg100"], p + "_a5m"), a5], p + "_a5w")
            att = n("Reshape", [a5, "sh_att4_" + tag], p + "_a4")
        pr = n("Softmax", [att], p + "_p", axis=-1)
        o = n("Transpose", [n("MatMul", [pr, v], p + "_pv")], p + "_pvt", perm=[0, 2, 1, 3])
        o = n("Reshape", [o, "sh_win" + tag], p + "_o")
        o = n("Add", [n("MatMul", [o, p + ".proj.w"], p + "_pm"), p + ".proj.b"], p + "_proj")
        t = n("Transpose", [n("Reshape", [o, "sh_merge" + tag], p + "_mg")], p + "_mgt", perm=[0, 1, 3, 2, 4, 5])
        xs = n("Reshape", [t, "sh_img" + tag], p + "_back")
        if shifted:  # roll by +shift
            xs = roll(roll(xs, p + "_uh", "s_back", None, "ax1", 1), p + "_uw", "s_back", None, "ax2", 2)
        o = n("Reshape", [xs, "sh_tok" + tag], p + "_tok")
        x = n("Add", [x, layer_norm(o, p + ".ln1", p + "_n1")], p + "_x1")
        h = n("Add", [n("MatMul", [x, p + ".fc1.w"], p + "_f1m"), p + ".fc1.b"], p + "_f1")
        if gelu_op:
            g = n("Gelu", [h], p + "_g")
        else:  # 0.5 * h * (1 + erf(h / sqrt(2)))
            e = n("Add", [n("Erf", [n("Div", [h, "sqrt2"], p + "_hd")], p + "_he"), "one"], p + "_he1")
            g = n("Mul", [n("Mul", [h, e], p + "_hm"), "half"], p + "_g")
        m = n("Add", [n("MatMul", [g, p + ".fc2.w"], p + "_f2m"), p + ".fc2.b"], p + "_f2")
        return n("Add", [x, layer_norm(m, p + ".ln2", p + "_n2")], p + "_x2")

    t = n("Conv", ["input", "pe.w", "pe.b"], "pe", strides=[S["patch"], S["patch"]])
    t = n("Transpose", [n("Reshape", [t, "sh_pe"], "pe_r")], "pe_t", perm=[0, 2, 1])
    x = layer_norm(t, "ln0", "x0")
    x = block(x, "b1", "1", False, False)
    x1 = block(x, "b2", "1", True, False)
    # patch merging: x[:, 0::2, 0::2, :], x[:, 1::2, 0::2, :], x[:, 0::2, 1::2, :], x[:, 1::2, 1::2, :] -> concat -> reduction -> norm
    # (the image view's target comes from a Shape sub-graph, as exports with a dynamic batch axis spell it)
    dyn = n("Concat", [n("Gather", [n("Shape", [x1], "x1_shape"), "ax0"], "x1_b", axis=0), "tail_hw_c"], "x1_imgshape", axis=0)
    img = n("Reshape", [x1, dyn], "pm_img")
    parts = [n("Slice", [img, st, "pm_end", "pm_ax", "pm_step"], "pm_" + st) for st in ("pm00", "pm10", "pm01", "pm11")]
    t = n("Reshape", [n("Concat", parts, "pm_cat", axis=-1), "sh_pm"], "pm_tok")
    x = layer_norm(n("MatMul", [t, "red.w"], "pm_red"), "lnm", "xm")
    x2 = block(x, "b3", "2", False, True)
    # a squeeze gate over the tokens of the deep stage
    gate = n("Sigmoid", [n("Mul", [n("ReduceSum", [x2, "ax1"], "x2_sum", keepdims=1), "inv_tokens"], "x2_mean")], "x2_gate")
    x2 = n("Mul", [x2, gate], "x2g")
    # DPT head
    f1 = n("Conv", [n("Reshape", [n("Transpose", [x1], "x1_t", perm=[0, 2, 1]), "sh_map1"], "x1_map"), "r1.w", "r1.b"], "f1")
    f2 = n("Conv", [n("Reshape", [n("Transpose", [x2], "x2_t", perm=[0, 2, 1]), "sh_map2"], "x2_map"), "r2.w", "r2.b"], "f2")
    up = n("Resize", [f2, "roi", "up_scales"], "f2_up", mode="linear", coordinate_transformation_mode="align_corners")
    m = n("Relu", [n("Conv", [n("Add", [f1, up], "fsum"), "fu.w", "fu.b"], "fuc", pads=[1, 1, 1, 1])], "fur")
    m = n("ConvTranspose", [m, "up1.w", "up1.b"], "u1", kernel_shape=[3, 3], strides=[2, 2], pads=[1, 1, 1, 1], output_padding=[1, 1])
    m = n("ConvTranspose", [m, "up2.w"], "u2", kernel_shape=[2, 2], strides=[2, 2])
    m = n("Relu", [n("Conv", [m, "h1.w", "h1.b"], "h1c", pads=[1, 1, 1, 1])], "h1r")
    m = n("Relu", [n("Conv", [m, "h2.w", "h2.b"], "h2c")], "h2r")
    n("Squeeze", [m], "output", axes=[1])
    for nd in out:  # roll() passed None placeholders for unused slots: none may survive
        assert all(i is not None for i in nd["in"]), nd
    return out


def spec_nodes(spec):
    """node dicts -> NodeProto payloads; an attribute given as bytes is an already encoded sub-graph (the body of a Loop / Scan)"""
    nodes = []
    for nd in spec:
        attrs = []
        for k, v in nd["attrs"].items():
            if isinstance(v, bytes):
                attrs.append(attr_graph(k, v))
            elif isinstance(v, str):
                attrs.append(attr_str(k, v))
            elif isinstance(v, float):
                attrs.append(attr_float(k, v))
            elif isinstance(v, int):
                attrs.append(attr_int(k, v))
            else:
                attrs.append(attr_ints(k, v))
        nodes.append(node(nd["op"], nd["in"], nd["out"], attrs, name=next(o for o in nd["out"] if o) + "_node"))
    return nodes


def spec_model(spec, floats, ints, in_dims, out_dims, extra_op=None):
    """any node-dict list as an ONNX file: float initialisers `floats`, int64 initialisers `ints` (name -> (values, dims or None))"""
    nodes = spec_nodes(spec)
    if extra_op:
        nodes.append(node(extra_op, ["output"], ["unused"], name="extra"))
    # the Where mask and names that start with "bool_" go out as ONNX bool (what masked_fill / loop conditions export), everything else as int64
    inits = [tensor(k, v) for k, v in floats.items()] + [(bool_tensor if k == "mask" or k.startswith("bool_") else int_tensor)(k, np.asarray(v[0]).reshape(-1), v[1]) for k, v in ints.items()]
    return model(nodes, inits, [value_info("input", 1, in_dims)], [value_info("output", 1, out_dims)])


# ---- a small image graph with a Loop (a residual block applied T times, a per-iteration scan output) and a Scan (a first-order recurrence down the
# rows), and its hand-unrolled twin: the control-flow ops against the same nodes written out (tests/test_depth_gpu.py) ----
LOOPNET = {"H": 16, "C": 4, "T": 3}


def loopnet_weights(seed):
    rng = np.random.default_rng(seed)
    f = lambda *sh: (rng.standard_normal(sh) * 0.3).astype(np.float32)  # noqa: E731
    H, Cc = LOOPNET["H"], LOOPNET["C"]
    return {"c0.w": f(Cc, 3, 3, 3), "c0.b": f(Cc), "cb.w": f(Cc, Cc, 3, 3), "cb.b": f(Cc), "c1.w": f(1, Cc, 3, 3), "c1.b": f(1),
            "decay": np.array([0.5], np.float32), "acc0": np.zeros((1, Cc, H), np.float32)}


def loopnet_ints():
    H, Cc, T = LOOPNET["H"], LOOPNET["C"], LOOPNET["T"]
    ints = {"shape_z": ([1, H, H], None), "shape_q": ([1, 1, H], None), "trip": ([T], []), "one": ([1], []), "limit": ([T], []), "bool_go": ([1], []),
            "shape_s": ([1, 1, H, H], None), "shape_a": ([1, 1, Cc, H], None), "shape_x": ([1, Cc, H], None), "ax0": ([0], None)}
    for t in range(H):
        ints["st%d" % t] = ([t], None)
        ints["en%d" % t] = ([t + 1], None)
    return ints


def _n(op, i, o, **a):
    return {"op": op, "in": i, "out": o, "attrs": a}


def _loop_body(y, sfx):
    """the block one iteration applies to `y` -> (nodes, new y, scan value)"""
    return [_n("Conv", [y, "cb.w", "cb.b"], ["t" + sfx], pads=[1, 1, 1, 1], kernel_shape=[3, 3]), _n("Relu", ["t" + sfx], ["r" + sfx]),
            _n("Add", ["r" + sfx, y], ["y" + sfx]), _n("ReduceSum", ["y" + sfx], ["s" + sfx], axes=[1], keepdims=0)], "y" + sfx, "s" + sfx


def _loopnet_tail(y_final, S, Y):
    return [_n("Conv", [y_final, "c1.w", "c1.b"], ["z"], pads=[1, 1, 1, 1], kernel_shape=[3, 3]), _n("Reshape", ["z", "shape_z"], ["zr"]),
            _n("ReduceSum", [S], ["m"], axes=[0], keepdims=0), _n("ReduceSum", [Y], ["q0"], axes=[0], keepdims=0),
            _n("ReduceSum", ["q0"], ["q1"], axes=[1], keepdims=0), _n("Reshape", ["q1", "shape_q"], ["qr"]),
            _n("Add", ["zr", "m"], ["zm"]), _n("Add", ["zm", "qr"], ["output"])]


def loopnet_unrolled_spec(reverse=False):
    H, T = LOOPNET["H"], LOOPNET["T"]
    spec = [_n("Conv", ["input", "c0.w", "c0.b"], ["y_0"], pads=[1, 1, 1, 1], kernel_shape=[3, 3])]
    y = "y_0"
    for t in range(T):
        nodes, y, s_ = _loop_body(y, "_%d" % (t + 1))
        spec += nodes + [_n("Reshape", [s_, "shape_s"], ["S%d" % t])]
    spec.append(_n("Concat", ["S%d" % t for t in range(T)], ["S"], axis=0))
    spec.append(_n("Transpose", [y, ], ["X"], perm=[2, 0, 1, 3]))
    acc = "acc0"
    for t in range(H):
        row = H - 1 - t if reverse else t                                   # a reversed Scan reads its slices last to first; its outputs stay in iteration order
        spec += [_n("Slice", ["X", "st%d" % row, "en%d" % row, "ax0"], ["xs%d" % t]), _n("Reshape", ["xs%d" % t, "shape_x"], ["x%d" % t]),
                 _n("Mul", [acc, "decay"], ["a%d" % t]), _n("Add", ["a%d" % t, "x%d" % t], ["acc%d" % (t + 1)]),
                 _n("Reshape", ["acc%d" % (t + 1), "shape_a"], ["Y%d" % t])]
        acc = "acc%d" % (t + 1)
    spec.append(_n("Concat", ["Y%d" % t for t in range(H)], ["Y"], axis=0))
    return spec + _loopnet_tail(y, "S", "Y")


def loopnet_spec(mode="count", reverse=False):
    """mode "count": trip count input, the body hands the condition through; "cond": no trip count, the body computes the condition from the
    iteration number (integer Add + Less on the host)"""
    H, Cc = LOOPNET["H"], LOOPNET["C"]
    body_nodes, y_out, s_out = _loop_body("y_in", "_b")
    if mode == "count":
        body_nodes.append(_n("Identity", ["go_in"], ["go_out"]))
    else:
        body_nodes += [_n("Add", ["it", "one"], ["it1"]), _n("Less", ["it1", "limit"], ["go_out"])]
    body = graph_proto(spec_nodes(body_nodes), [], [value_info("it", 7, []), value_info("go_in", 9, []), value_info("y_in", 1, [1, Cc, H, H])],
                       [value_info("go_out", 9, []), value_info(y_out, 1, [1, Cc, H, H]), value_info(s_out, 1, [1, H, H])], name=b"loop_body")
    scan_nodes = [_n("Mul", ["acc_in", "decay"], ["a_b"]), _n("Add", ["a_b", "x_in"], ["acc_out"]), _n("Identity", ["acc_out"], ["y_row"])]
    scan_body = graph_proto(spec_nodes(scan_nodes), [], [value_info("acc_in", 1, [1, Cc, H]), value_info("x_in", 1, [1, Cc, H])],
                            [value_info("acc_out", 1, [1, Cc, H]), value_info("y_row", 1, [1, Cc, H])], name=b"scan_body")
    return [_n("Conv", ["input", "c0.w", "c0.b"], ["y_0"], pads=[1, 1, 1, 1], kernel_shape=[3, 3]),
            _n("Loop", ["trip" if mode == "count" else "", "bool_go", "y_0"], ["y_fin", "S"], body=body),
            _n("Transpose", ["y_fin"], ["X"], perm=[2, 0, 1, 3]),
            _n("Scan", ["acc0", "X"], ["acc_fin", "Y"], body=scan_body, num_scan_inputs=1, **({"scan_input_directions": [1]} if reverse else {}))] + _loopnet_tail("y_fin", "S", "Y")


def loopnet_model(W, spec):
    H = LOOPNET["H"]
    return spec_model(spec, W, loopnet_ints(), [1, 3, H, H], [1, H, H])


def swin_model(W, extra_op=None):
    H = SWIN["H"]
    return spec_model(swin_spec(), W, swin_ints(), [1, 3, H, H], [1, H, H], extra_op)


def swin_consts(W):
    """every constant by name as ndarrays, for the numpy oracle and the torch script"""
    c = dict(W)
    for k, (vals, dims) in swin_ints().items():
        a = np.asarray(vals, np.int64)
        c[k] = a.reshape(dims) if dims is not None else a
    return c


# ---- a detector of the YOLOv5u class (the file the reference names: yolov5nu.onnx, src/cortex/tk_cortex_main.h:71, tests/tk_cortex_test.cpp:41):
# C3 blocks + SPPF backbone, PAN head, the anchor-free decoupled Detect head with DFL — spelled the way an Ultralytics export spells it: every
# Conv module = Conv (BN folded) -> Sigmoid -> Mul, Upsample = Resize(nearest, x2), Detect = per scale Concat(box, cls) -> Reshape -> Concat
# over scales -> Split -> [Reshape, Transpose, Softmax(axis 1), Conv(arange 16), Reshape] -> Slice / Sub / Add / Div / Concat (dist2bbox, xywh)
# -> Mul(strides) and Sigmoid(cls) -> Concat -> [1, 4 + nc, anchors].  Channel widths are a fraction of the nano model's: no checkpoint is
# available offline, the graph class is what is tested. ----

YOLO5_CH = (8, 16, 32, 64, 128)   # c1 .. c5 (the nano model: 16, 32, 64, 128, 256)
YOLO5_DEPTH = (1, 2, 3, 1)        # bottlenecks of the backbone's four C3 blocks (nano: 1, 2, 3, 1)


def yolo5_modules(nc):
    """(module path, c_in, c_out, k, stride, activation) of every convolution, in no particular order"""
    k1, k2, k3, k4, k5 = YOLO5_CH
    out = []

    def conv(name, ci, co, k=1, s=1, act=True):
        out.append((name, ci, co, k, s, act))

    def c3(name, ci, co, n):
        h = co // 2
        conv(name + ".cv1", ci, h); conv(name + ".cv2", ci, h); conv(name + ".cv3", 2 * h, co)
        for i in range(n):
            conv("%s.m%d.cv1" % (name, i), h, h, 1); conv("%s.m%d.cv2" % (name, i), h, h, 3)

    conv("b0", 3, k1, 6, 2); conv("b1", k1, k2, 3, 2); c3("b2", k2, k2, YOLO5_DEPTH[0]); conv("b3", k2, k3, 3, 2); c3("b4", k3, k3, YOLO5_DEPTH[1])
    conv("b5", k3, k4, 3, 2); c3("b6", k4, k4, YOLO5_DEPTH[2]); conv("b7", k4, k5, 3, 2); c3("b8", k5, k5, YOLO5_DEPTH[3])
    conv("b9.cv1", k5, k5 // 2); conv("b9.cv2", 2 * k5, k5)
    conv("h10", k5, k4); c3("h13", 2 * k4, k4, 1); conv("h14", k4, k3); c3("h17", 2 * k3, k3, 1)
    conv("h18", k3, k3, 3, 2); c3("h20", 2 * k3, k4, 1); conv("h21", k4, k4, 3, 2); c3("h23", 2 * k4, k5, 1)
    cb, cc = 16, max(k3, 16)
    for i, ch in enumerate((k3, k4, k5)):
        conv("d.cv2.%d.0" % i, ch, cb, 3); conv("d.cv2.%d.1" % i, cb, cb, 3); conv("d.cv2.%d.2" % i, cb, 64, 1, 1, False)
        conv("d.cv3.%d.0" % i, ch, cc, 3); conv("d.cv3.%d.1" % i, cc, cc, 3); conv("d.cv3.%d.2" % i, cc, nc, 1, 1, False)
    return out


def yolo5_weights(seed, nc=80, cls_bias=-1.5):
    rng = np.random.default_rng(seed)
    W = {}
    for name, ci, co, k, s, act in yolo5_modules(nc):
        W[name + ".w"] = (rng.standard_normal((co, ci, k, k)) * 1.4 / np.sqrt(ci * k * k)).astype(np.float32)
        W[name + ".b"] = (0.05 * rng.standard_normal(co)).astype(np.float32)
        if name.startswith("d.cv3.") and name.endswith(".2"):
            W[name + ".b"] = (W[name + ".b"] + cls_bias).astype(np.float32)
    return W


def yolo5_anchors(H, W):
    """anchor centres [2, A] in grid units and strides [1, A], scale after scale (Ultralytics make_anchors, offset 0.5)"""
    pts, strides = [], []
    for st in (8, 16, 32):
        h, w = H // st, W // st
        ys, xs = np.meshgrid(np.arange(h, dtype=np.float32) + 0.5, np.arange(w, dtype=np.float32) + 0.5, indexing="ij")
        pts.append(np.stack([xs.reshape(-1), ys.reshape(-1)], 0))
        strides.append(np.full((1, h * w), float(st), np.float32))
    return np.concatenate(pts, 1).astype(np.float32), np.concatenate(strides, 1)


def yolo5_spec(nc, H, W):
    """(nodes, float constants beside the weights, int64 constants) of the export"""
    spec = []
    floats, ints = {}, {}
    mods = {m[0]: m for m in yolo5_modules(nc)}

    def n(op, i, o, **attrs):
        spec.append({"op": op, "in": i, "out": o if isinstance(o, list) else [o], "attrs": attrs})
        return o

    def conv(name, x):
        _, ci, co, k, s, act = mods[name]
        p = k // 2 if k != 6 else 2
        y = n("Conv", [x, name + ".w", name + ".b"], name + "/c", strides=[s, s], pads=[p, p, p, p], kernel_shape=[k, k])
        if not act:
            return y
        return n("Mul", [y, n("Sigmoid", [y], name + "/s")], name + "/y")

    def c3(name, x, nb, shortcut):
        a = conv(name + ".cv1", x)
        for i in range(nb):
            t = conv("%s.m%d.cv2" % (name, i), conv("%s.m%d.cv1" % (name, i), a))
            a = n("Add", [a, t], "%s.m%d/add" % (name, i)) if shortcut else t
        b = conv(name + ".cv2", x)
        return conv(name + ".cv3", n("Concat", [a, b], name + "/cat", axis=1))

    def up(x, name):
        return n("Resize", [x, "roi", "up_scales"], name, mode="nearest", coordinate_transformation_mode="asymmetric", nearest_mode="floor")

    floats["roi"] = np.zeros(0, np.float32)
    floats["up_scales"] = np.array([1, 1, 2, 2], np.float32)
    x = conv("b1", conv("b0", "images"))
    x = c3("b2", x, YOLO5_DEPTH[0], True)
    p3 = c3("b4", conv("b3", x), YOLO5_DEPTH[1], True)
    p4 = c3("b6", conv("b5", p3), YOLO5_DEPTH[2], True)
    x = c3("b8", conv("b7", p4), YOLO5_DEPTH[3], True)
    y0 = conv("b9.cv1", x)
    mp = dict(kernel_shape=[5, 5], strides=[1, 1], pads=[2, 2, 2, 2])
    y1 = n("MaxPool", [y0], "b9/mp1", **mp); y2 = n("MaxPool", [y1], "b9/mp2", **mp); y3 = n("MaxPool", [y2], "b9/mp3", **mp)
    x9 = conv("b9.cv2", n("Concat", [y0, y1, y2, y3], "b9/cat", axis=1))
    h10 = conv("h10", x9)
    h13 = c3("h13", n("Concat", [up(h10, "h11"), p4], "h12", axis=1), 1, False)
    h14 = conv("h14", h13)
    h17 = c3("h17", n("Concat", [up(h14, "h15"), p3], "h16", axis=1), 1, False)
    h20 = c3("h20", n("Concat", [conv("h18", h17), h14], "h19", axis=1), 1, False)
    h23 = c3("h23", n("Concat", [conv("h21", h20), h10], "h22", axis=1), 1, False)
    flat = []
    for i, f in enumerate((h17, h20, h23)):
        box = conv("d.cv2.%d.2" % i, conv("d.cv2.%d.1" % i, conv("d.cv2.%d.0" % i, f)))
        cls = conv("d.cv3.%d.2" % i, conv("d.cv3.%d.1" % i, conv("d.cv3.%d.0" % i, f)))
        cat = n("Concat", [box, cls], "d/cat%d" % i, axis=1)
        ints["d/shape%d" % i] = ([1, 64 + nc, -1], None)
        flat.append(n("Reshape", [cat, "d/shape%d" % i], "d/flat%d" % i))
    allc = n("Concat", flat, "d/all", axis=2)
    ints["d/split"] = ([64, nc], None)
    n("Split", [allc, "d/split"], ["d/box", "d/cls"], axis=1)
    ints["d/dfl_shape"] = ([1, 4, 16, -1], None)
    t = n("Transpose", [n("Reshape", ["d/box", "d/dfl_shape"], "d/dfl_r")], "d/dfl_t", perm=[0, 2, 1, 3])
    t = n("Conv", [n("Softmax", [t], "d/dfl_s", axis=1), "d/dfl_w"], "d/dfl_c")
    floats["d/dfl_w"] = np.arange(16, dtype=np.float32).reshape(1, 16, 1, 1)
    ints["d/dist_shape"] = ([1, 4, -1], None)
    dist = n("Reshape", [t, "d/dist_shape"], "d/dist")
    for nm, v in (("k0", [0]), ("k2", [2]), ("k4", [4]), ("ax1", [1])):
        ints["d/" + nm] = (v, None)
    lt = n("Slice", [dist, "d/k0", "d/k2", "d/ax1"], "d/lt")
    rb = n("Slice", [dist, "d/k2", "d/k4", "d/ax1"], "d/rb")
    pts, strides = yolo5_anchors(H, W)
    floats["d/anchors"] = pts[None]
    floats["d/strides"] = strides
    floats["d/two"] = np.array(2.0, np.float32)
    x1y1 = n("Sub", ["d/anchors", lt], "d/x1y1")
    x2y2 = n("Add", ["d/anchors", rb], "d/x2y2")
    cxy = n("Div", [n("Add", [x1y1, x2y2], "d/sum"), "d/two"], "d/cxy")
    wh = n("Sub", [x2y2, x1y1], "d/wh")
    dbox = n("Mul", [n("Concat", [cxy, wh], "d/xywh", axis=1), "d/strides"], "d/dbox")
    n("Concat", [dbox, n("Sigmoid", ["d/cls"], "d/prob")], "output", axis=1)
    return spec, floats, ints


def yolo5_model(W, nc, H, Wd, extra_op=None):
    spec, floats, ints = yolo5_spec(nc, H, Wd)
    fl = dict(W)
    fl.update(floats)
    nodes = []
    for nd in spec:
        attrs = []
        for k, v in nd["attrs"].items():
            attrs.append(attr_str(k, v) if isinstance(v, str) else attr_float(k, v) if isinstance(v, float) else attr_int(k, v) if isinstance(v, int) else attr_ints(k, v))
        nodes.append(node(nd["op"], nd["in"], nd["out"], attrs, name=nd["out"][0] + "_node"))
    if extra_op:
        nodes.append(node(extra_op, ["output"], ["unused"], name="extra"))
    inits = [tensor(k, v) for k, v in fl.items()] + [int_tensor(k, np.asarray(v[0]).reshape(-1), v[1]) for k, v in ints.items()]
    A = (H // 8) * (Wd // 8) + (H // 16) * (Wd // 16) + (H // 32) * (Wd // 32)
    return model(nodes, inits, [value_info("images", 1, [1, 3, H, Wd])], [value_info("output", 1, [1, 4 + nc, A])])
