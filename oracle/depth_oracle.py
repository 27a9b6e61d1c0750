"""CPU restatement (numpy, float32) of the depth path — TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product.

What it restates
  * the ONNX node semantics the depth network needs (the reference hands the file to ONNX Runtime, src/vision/tk_depth_midas.c:397-440;
    ONNX operator specification, opset 11-17): Conv (groups, strides, pads, dilations), BatchNormalization, Clip, Relu, LeakyRelu, Sigmoid,
    Add / Mul with broadcasting, Concat, MaxPool, AveragePool, GlobalAveragePool, Pad (constant), Resize (linear / nearest with the
    half_pixel, pytorch_half_pixel, align_corners and asymmetric coordinate modes), Squeeze; and the token-sequence ops of the DPT / Swin class
    (the model the reference names, tk_depth_midas.c:8): LayerNormalization, Erf, Gelu, MatMul, Softmax, Transpose, Reshape, Gather, strided
    Slice, ReduceL2 / ReduceSum / ReduceMax / ReduceMin, Max / Min / Where, Expand, Shape, ConvTranspose (_seq_op).  A graph is a list of node dicts
    {"op", "in", "out", "attrs"} (tests/onnx_util.depth_spec builds one and writes the same list as an ONNX file).
  * convert_inverse_depth_to_metric (src/vision/tk_depth_midas.c:471-499), bit for bit in float32.
  * calculate_raw_distance and fuse_object_and_depth_data (src/vision/src/object_analysis.rs:134-279), with trackers in creation order.
Pinning: the network semantics against torch on a seeded graph (tests/golden/depth_net.npz, made by tests/golden/make_depth_golden.py);
the fusion against the reference's own unit test vector (object_analysis.rs:286-347: 10 m, then 12 m -> between 10 and 12).
Convolutions accumulate in float32 in (input channel, kernel row, kernel column) order with separate multiply and add, the GPU uses one
fma chain in the same order: agreement is to rounding (tests allow 2e-5 relative to the tensor's scale), not bit-exact.
"""
import math

import numpy as np

F = np.float32


def conv2d(x, w, b, strides=(1, 1), pads=(0, 0, 0, 0), dilations=(1, 1), group=1):
    n, c, h, wd = x.shape
    m, cg, kh, kw = w.shape
    sh, sw = strides
    dh, dw = dilations
    pt, pl, pb, pr = pads
    ho = (h + pt + pb - dh * (kh - 1) - 1) // sh + 1
    wo = (wd + pl + pr - dw * (kw - 1) - 1) // sw + 1
    xp = np.zeros((n, c, h + pt + pb, wd + pl + pr), F)
    xp[:, :, pt:pt + h, pl:pl + wd] = x
    y = np.zeros((n, m, ho, wo), F)
    mg = m // group
    for g in range(group):
        for ci in range(cg):
            for a in range(kh):
                for q in range(kw):
                    patch = xp[:, g * cg + ci, a * dh:a * dh + (ho - 1) * sh + 1:sh, q * dw:q * dw + (wo - 1) * sw + 1:sw]  # [n, ho, wo]
                    wv = w[g * mg:(g + 1) * mg, ci, a, q]  # [mg]
                    y[:, g * mg:(g + 1) * mg] = (y[:, g * mg:(g + 1) * mg] + patch[:, None, :, :] * wv[None, :, None, None]).astype(F)
    if b is not None:
        y = (y + b.reshape(1, -1, 1, 1)).astype(F)
    return y


def _src(o, n_in, n_out, scale, mode):
    o = np.arange(n_out, dtype=F) if o is None else o
    if mode == "align_corners":
        return (o * F(n_in - 1) / F(n_out - 1)).astype(F) if n_out > 1 else np.zeros(n_out, F)
    if mode == "asymmetric":
        return (o / F(scale)).astype(F)
    if mode == "pytorch_half_pixel" and n_out <= 1:
        return np.zeros(n_out, F)
    return ((o + F(0.5)) / F(scale) - F(0.5)).astype(F)


def resize(x, ho, wo, sh, sw, mode="linear", ct="half_pixel", nearest_mode="round_prefer_floor"):
    n, c, h, w = x.shape
    fy, fx = _src(None, h, ho, sh, ct), _src(None, w, wo, sw, ct)
    if mode == "nearest":
        def near(f, lim):
            r = {"floor": np.floor(f), "ceil": np.ceil(f), "round_prefer_ceil": np.floor(f + F(0.5))}.get(nearest_mode, np.ceil(f - F(0.5)))
            return np.clip(r.astype(np.int64), 0, lim - 1)
        return x[:, :, near(fy, h)][:, :, :, near(fx, w)].astype(F)
    cy, cx = np.clip(fy, 0, h - 1).astype(F), np.clip(fx, 0, w - 1).astype(F)
    y0, x0 = np.floor(cy).astype(np.int64), np.floor(cx).astype(np.int64)
    y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
    dy, dx = (cy - y0.astype(F)).astype(F)[None, None, :, None], (cx - x0.astype(F)).astype(F)[None, None, None, :]
    v00, v01 = x[:, :, y0][:, :, :, x0], x[:, :, y0][:, :, :, x1]
    v10, v11 = x[:, :, y1][:, :, :, x0], x[:, :, y1][:, :, :, x1]
    one = F(1.0)
    top = ((one - dx) * v00 + dx * v01).astype(F)
    bot = ((one - dx) * v10 + dx * v11).astype(F)
    return ((one - dy) * top + dy * bot).astype(F)


def pool(x, kernel, strides, pads, is_max, count_include_pad=0):
    n, c, h, w = x.shape
    kh, kw = kernel
    sh, sw = strides
    pt, pl, pb, pr = pads
    ho = (h + pt + pb - kh) // sh + 1
    wo = (w + pl + pr - kw) // sw + 1
    y = np.zeros((n, c, ho, wo), F)
    for i in range(ho):
        for j in range(wo):
            h0, w0 = i * sh - pt, j * sw - pl
            hs, he, ws, we = max(h0, 0), min(h0 + kh, h), max(w0, 0), min(w0 + kw, w)
            win = x[:, :, hs:he, ws:we]
            if is_max:
                y[:, :, i, j] = win.max(axis=(2, 3))
            else:
                acc = np.zeros((n, c), F)
                for a in range(hs, he):  # row-major accumulation, as the kernel's two loops
                    for q in range(ws, we):
                        acc = (acc + x[:, :, a, q]).astype(F)
                cnt = kh * kw if count_include_pad else (he - hs) * (we - ws)
                y[:, :, i, j] = (acc / F(cnt)).astype(F)
    return y


def run_graph(spec, consts, feeds):
    """spec: node dicts in execution order; consts / feeds: name -> ndarray.  Returns every tensor by name."""
    v = dict(consts)
    v.update(feeds)
    for nd in spec:
        op, i, o, a = nd["op"], nd["in"], nd["out"], nd.get("attrs", {})
        x = v[i[0]] if i and i[0] else None
        if op == "Conv":
            pads = a.get("pads", [0, 0, 0, 0])
            y = conv2d(x, v[i[1]], v[i[2]] if len(i) > 2 and i[2] else None, a.get("strides", [1, 1]), pads, a.get("dilations", [1, 1]), a.get("group", 1))
        elif op == "BatchNormalization":
            sc, bi, mu, va = (v[k].reshape(1, -1, 1, 1) for k in i[1:5])
            y = (((x - mu) / np.sqrt((va + F(a.get("epsilon", 1e-5))).astype(F))).astype(F) * sc + bi).astype(F)
        elif op == "Clip":
            lo = F(a["min"]) if "min" in a else (v[i[1]].reshape(()) if len(i) > 1 and i[1] else F(-np.inf))
            hi = F(a["max"]) if "max" in a else (v[i[2]].reshape(()) if len(i) > 2 and i[2] else F(np.inf))
            y = np.minimum(np.maximum(x, lo), hi).astype(F)
        elif op == "Relu":
            y = np.maximum(x, F(0))
        elif op == "LeakyRelu":
            y = np.where(x >= 0, x, x * F(a.get("alpha", 0.01))).astype(F)
        elif op == "Sigmoid":
            y = (F(1) / (F(1) + np.exp(-x.astype(np.float64)))).astype(F)
        elif op == "Add":
            y = (x + v[i[1]]).astype(F)
        elif op == "Mul":
            y = (x * v[i[1]]).astype(F)
        elif op == "Concat":
            y = np.concatenate([np.atleast_1d(v[k]) for k in i], axis=a.get("axis", 1))
        elif op == "MaxPool" or op == "AveragePool":
            y = pool(x, a["kernel_shape"], a.get("strides", [1, 1]), a.get("pads", [0, 0, 0, 0]), op == "MaxPool", a.get("count_include_pad", 0))
        elif op == "GlobalAveragePool":
            y = pool(x, x.shape[2:], [1, 1], [0, 0, 0, 0], False)
        elif op == "Pad":
            p = [int(t) for t in v[i[1]]]
            r = x.ndim
            y = np.pad(x, [(p[d], p[r + d]) for d in range(r)], mode="constant").astype(F)
        elif op == "Resize":
            sc = v[i[2]]
            ho, wo = int(math.floor(x.shape[2] * float(sc[2]))), int(math.floor(x.shape[3] * float(sc[3])))
            y = resize(x, ho, wo, F(sc[2]), F(sc[3]), a.get("mode", "nearest"), a.get("coordinate_transformation_mode", "half_pixel"), a.get("nearest_mode", "round_prefer_floor"))
        elif op == "Squeeze":
            y = np.squeeze(x, axis=tuple(a["axes"]))
        else:
            y = _seq_op(op, i, a, v)
        v[o[0]] = np.ascontiguousarray(y, np.int64 if y.dtype.kind in "iub" else F)
    return v


def conv_transpose2d(x, w, b, strides, pads, output_padding):
    """ONNX ConvTranspose, group 1: y[n, co, iy * sh - pt + ky, ix * sw - pl + kx] += x[n, ci, iy, ix] * w[ci, co, ky, kx]"""
    n, ci, h, wd = x.shape
    _, co, kh, kw = w.shape
    sh, sw = strides
    pt, pl, pb, pr = pads
    ho = (h - 1) * sh + kh - pt - pb + output_padding[0]
    wo = (wd - 1) * sw + kw - pl - pr + output_padding[1]
    full = np.zeros((n, co, (h - 1) * sh + kh + output_padding[0], (wd - 1) * sw + kw + output_padding[1]), F)
    for c in range(ci):
        for ky in range(kh):
            for kx in range(kw):
                full[:, :, ky:ky + (h - 1) * sh + 1:sh, kx:kx + (wd - 1) * sw + 1:sw] += (x[:, c, None, :, :] * w[c, :, ky, kx][None, :, None, None]).astype(F)
    y = full[:, :, pt:pt + ho, pl:pl + wo]
    return (y + b.reshape(1, -1, 1, 1)).astype(F) if b is not None else y.astype(F)


def _seq_op(op, i, a, v):
    """the token-sequence ops of the DPT / Swin graph class (ONNX operator specification, opsets 13-20)"""
    from scipy.special import erf
    x = v[i[0]] if i and i[0] else None
    if op == "LayerNormalization":
        mu = x.mean(axis=-1, keepdims=True, dtype=F)
        d = (x - mu).astype(F)
        var = (d * d).mean(axis=-1, keepdims=True, dtype=F)
        y = (d / np.sqrt((var + F(a.get("epsilon", 1e-5))).astype(F))).astype(F) * v[i[1]]
        return (y + v[i[2]]).astype(F) if len(i) > 2 and i[2] else y.astype(F)
    if op == "Erf":
        return erf(x.astype(np.float64)).astype(F)
    if op == "Gelu":
        x64 = x.astype(np.float64)
        if a.get("approximate", "none") == "tanh":
            return (0.5 * x64 * (1 + np.tanh(math.sqrt(2 / math.pi) * (x64 + 0.044715 * x64 ** 3)))).astype(F)
        return (0.5 * x64 * (1 + erf(x64 / math.sqrt(2.0)))).astype(F)
    if op == "MatMul":
        return np.matmul(x, v[i[1]]).astype(F)
    if op == "Softmax":
        e = np.exp((x - x.max(axis=a.get("axis", -1), keepdims=True)).astype(np.float64))
        return (e / e.sum(axis=a.get("axis", -1), keepdims=True)).astype(F)
    if op == "Transpose":
        return np.transpose(x, a["perm"])
    if op == "Reshape":
        return x.reshape([int(t) for t in v[i[1]]])
    if op == "Unsqueeze":
        return np.expand_dims(x, tuple(int(t) for t in np.asarray(v[i[1]]).reshape(-1)))
    if op == "Shape":
        return np.array(x.shape, np.int64)
    if op == "Gather":
        return np.take(x, np.asarray(v[i[1]], np.int64), axis=a.get("axis", 0))
    if op == "Slice":
        st, en = np.asarray(v[i[1]]).reshape(-1), np.asarray(v[i[2]]).reshape(-1)
        ax = np.asarray(v[i[3]]).reshape(-1) if len(i) > 3 and i[3] else np.arange(len(st))
        sp = np.asarray(v[i[4]]).reshape(-1) if len(i) > 4 and i[4] else np.ones(len(st), np.int64)
        sl = [slice(None)] * x.ndim
        for s_, e_, a_, p_ in zip(st, en, ax, sp):
            sl[int(a_)] = slice(int(s_), int(e_), int(p_))
        return x[tuple(sl)]
    if op == "Div":
        return (x / v[i[1]]).astype(F)
    if op == "Max":
        return np.maximum(x, v[i[1]]).astype(F)
    if op == "Min":
        return np.minimum(x, v[i[1]]).astype(F)
    if op == "Where":
        return np.where(np.asarray(v[i[0]]) != 0, v[i[1]], v[i[2]]).astype(F)
    if op == "Expand":
        return np.broadcast_to(x, np.broadcast_shapes(x.shape, tuple(int(t) for t in v[i[1]]))).astype(F)
    if op in ("ReduceL2", "ReduceSum", "ReduceMax", "ReduceMin"):
        axes = a["axes"] if "axes" in a else [int(t) for t in np.asarray(v[i[1]]).reshape(-1)]
        keep = bool(a.get("keepdims", 1))
        if op == "ReduceL2":
            return np.sqrt((x.astype(np.float64) ** 2).sum(axis=tuple(axes), keepdims=keep)).astype(F)
        fn = {"ReduceSum": np.sum, "ReduceMax": np.max, "ReduceMin": np.min}[op]
        return fn(x.astype(np.float64), axis=tuple(axes), keepdims=keep).astype(F)
    if op == "ConvTranspose":
        return conv_transpose2d(x, v[i[1]], v[i[2]] if len(i) > 2 and i[2] else None, a.get("strides", [1, 1]), a.get("pads", [0, 0, 0, 0]), a.get("output_padding", [0, 0]))
    raise NotImplementedError(op)


def to_metric(raw, min_depth=0.1, max_depth=10.0):
    """convert_inverse_depth_to_metric, src/vision/tk_depth_midas.c:471-499"""
    raw = np.asarray(raw, F)
    lo, hi = raw.min(), raw.max()
    if F(hi - lo) < F(1e-6):
        return np.full(raw.shape, F(max_depth), F)
    normalized = ((raw - lo) / F(hi - lo)).astype(F)
    return (F(max_depth) - normalized * F(F(max_depth) - F(min_depth))).astype(F)


def _round_half_away(v):
    return math.floor(v + 0.5) if v >= 0 else -math.floor(-v + 0.5)


def raw_distance(box, depth, frame_w, frame_h):
    """calculate_raw_distance, object_analysis.rs:227-279.  box = (x, y, w, h) ints; depth [dh][dw] float32"""
    dh, dw = depth.shape
    x, y, w, h = box
    n = [F(F(x) / F(frame_w)), F(F(y) / F(frame_h)), F(F(x + w) / F(frame_w)), F(F(y + h) / F(frame_h))]
    def sat(t):
        t = _round_half_away(float(t))
        return 0 if t < 0 else min(int(t), 0xFFFFFFFF)
    x0, y0 = sat(F(n[0] * F(dw - 1))), sat(F(n[1] * F(dh - 1)))
    x1, y1 = sat(F(n[2] * F(dw - 1))), sat(F(n[3] * F(dh - 1)))
    if x0 >= dw or y0 >= dh or x1 >= dw or y1 >= dh or x0 >= x1 or y0 >= y1:
        return F(-1.0)
    win = depth[y0:y1 + 1, x0:x1 + 1].reshape(-1)
    vals = np.sort(win[(win > F(0.1)) & (win < F(100.0))])
    if len(vals) < 10:
        return F(-1.0)
    q1, q3 = vals[len(vals) // 4], vals[len(vals) * 3 // 4]
    iqr = F(q3 - q1)
    lo, hi = F(q1 - F(1.5) * iqr), F(q3 + F(1.5) * iqr)
    keep = vals[(vals >= lo) & (vals <= hi)]
    if len(keep) == 0:
        return F(-1.0)
    s = F(0)
    for d in keep:
        s = F(s + d)
    return F(s / F(len(keep)))


def iou(a, b):
    xl, yt = max(a[0], b[0]), max(a[1], b[1])
    xr, yb = min(a[0] + a[2], b[0] + b[2]), min(a[1] + a[3], b[1] + b[3])
    if xr < xl or yb < yt:
        return F(0)
    inter = F(F(xr - xl) * F(yb - yt))
    uni = F(F(F(a[2] * a[3]) + F(b[2] * b[3])) - inter)
    return F(inter / uni) if uni > 0 else F(0)


class Fusion:
    """fuse_object_and_depth_data, object_analysis.rs:134-223; one result per detection (None: no valid depth under the box)"""

    def __init__(self):
        self.trackers = []  # dicts: cls, box, x, p, unseen

    def fuse(self, boxes, classes, depth, frame_w, frame_h, fx, fy):
        for t in self.trackers:
            t["matched"] = False
        served = [None] * len(boxes)
        for i, b in enumerate(boxes):
            raw = raw_distance(b, depth, frame_w, frame_h)
            if raw < 0:
                continue
            best, best_iou = None, F(0)
            for t in self.trackers:
                v = iou(b, t["box"])
                if v > F(0.4) and v > best_iou:
                    best, best_iou = t, v
            if best is not None:
                best["p"] = F(best["p"] + F(0.1))
                k = F(best["p"] / F(best["p"] + F(0.5)))
                best["x"] = F(best["x"] + F(k * F(raw - best["x"])))
                best["p"] = F(F(F(1.0) - k) * best["p"])
                best["box"], best["unseen"], best["matched"] = tuple(b), 0, True
                served[i] = best
            else:
                t = {"cls": classes[i], "box": tuple(b), "x": F(raw), "p": F(1.0), "unseen": 0, "matched": True}
                self.trackers.append(t)
                served[i] = t
        out = []
        for i, t in enumerate(served):
            if t is None:
                out.append(None)
                continue
            d = t["x"]
            if d > 0:
                out.append((d, F(F(F(t["box"][2]) * d) / F(fx)), F(F(F(t["box"][3]) * d) / F(fy))))
            else:
                out.append((d, F(-1), F(-1)))
        keep = []
        for t in self.trackers:
            if not t["matched"]:
                t["unseen"] += 1
                if t["unseen"] > 5:
                    continue
            keep.append(t)
        self.trackers = keep
        return out
