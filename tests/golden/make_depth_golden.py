"""Pins the depth-network semantics to torch: evaluates tests/onnx_util.depth_spec() (convolutional MiDaS class) and
tests/onnx_util.swin_spec() (DPT / Swin-transformer class) with torch on seeded weights and a seeded input, and stores input, weights'
seed and output in tests/golden/depth_net.npz and tests/golden/depth_swin.npz.  Run here (torch is in the image):
    python tests/golden/make_depth_golden.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as TF

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import onnx_util as OX  # noqa: E402

SEED, H, W = 11, 64, 64


def run_torch(spec, consts, x):
    v = {k: torch.from_numpy(np.asarray(a)) for k, a in consts.items()}
    v["input"] = torch.from_numpy(x)
    for nd in spec:
        op, i, o, a = nd["op"], nd["in"], nd["out"][0], nd["attrs"]
        t = v[i[0]]
        if op == "Conv":
            p = a.get("pads", [0, 0, 0, 0])
            assert p[0] == p[2] and p[1] == p[3]
            y = TF.conv2d(t, v[i[1]], v[i[2]], stride=a.get("strides", [1, 1]), padding=(p[0], p[1]), groups=a.get("group", 1))
        elif op == "BatchNormalization":
            y = TF.batch_norm(t, v[i[3]], v[i[4]], v[i[1]], v[i[2]], training=False, eps=a["epsilon"])
        elif op == "Clip":
            lo = a["min"] if "min" in a else float(v[i[1]])
            hi = a["max"] if "max" in a else (float(v[i[2]]) if len(i) > 2 else float("inf"))
            y = torch.clamp(t, lo, hi)
        elif op == "Relu":
            y = torch.relu(t)
        elif op == "LeakyRelu":
            y = TF.leaky_relu(t, a["alpha"])
        elif op == "Sigmoid":
            y = torch.sigmoid(t)
        elif op == "Add":
            y = t + v[i[1]]
        elif op == "Mul":
            y = t * v[i[1]]
        elif op == "Concat":
            y = torch.cat([v[k] for k in i], dim=a["axis"])
        elif op == "MaxPool":
            y = TF.max_pool2d(t, a["kernel_shape"], a["strides"], padding=(a["pads"][0], a["pads"][1]))
        elif op == "AveragePool":
            y = TF.avg_pool2d(t, a["kernel_shape"], a["strides"], padding=(a["pads"][0], a["pads"][1]), count_include_pad=bool(a["count_include_pad"]))
        elif op == "GlobalAveragePool":
            y = TF.adaptive_avg_pool2d(t, 1)
        elif op == "Resize":
            y = TF.interpolate(t, scale_factor=2, mode="bilinear", align_corners=a["coordinate_transformation_mode"] == "align_corners")
        elif op == "Pad":
            p = [int(q) for q in v[i[1]]]
            y = TF.pad(t, (p[3], p[7], p[2], p[6]))
        elif op == "Squeeze":
            y = t.squeeze(a["axes"][0])
        # ---- token-sequence ops (DPT / Swin class): the torch call each ONNX op is the export of ----
        elif op == "LayerNormalization":
            y = TF.layer_norm(t, (t.shape[-1],), v[i[1]], v[i[2]] if len(i) > 2 else None, a["epsilon"])
        elif op == "Erf":
            y = torch.erf(t)
        elif op == "Gelu":
            y = TF.gelu(t, approximate=a.get("approximate", "none"))
        elif op == "MatMul":
            y = torch.matmul(t, v[i[1]])
        elif op == "Softmax":
            y = torch.softmax(t, dim=a["axis"])
        elif op == "Transpose":
            y = t.permute(*a["perm"])
        elif op == "Reshape":
            y = t.reshape([int(q) for q in v[i[1]]])
        elif op == "Unsqueeze":
            y = t.unsqueeze(int(v[i[1]].reshape(-1)[0]))
        elif op == "Shape":
            y = torch.tensor(list(t.shape), dtype=torch.int64)
        elif op == "Gather":
            idx = v[i[1]]
            y = torch.index_select(t, a["axis"], idx.reshape(-1)).reshape(list(t.shape[:a["axis"]]) + list(idx.shape) + list(t.shape[a["axis"] + 1:]))
        elif op == "Slice":
            st, en, ax = v[i[1]].reshape(-1), v[i[2]].reshape(-1), v[i[3]].reshape(-1)
            sp = v[i[4]].reshape(-1) if len(i) > 4 else torch.ones_like(st)
            sl = [slice(None)] * t.dim()
            for s_, e_, a_, p_ in zip(st, en, ax, sp):
                sl[int(a_)] = slice(int(s_), int(e_), int(p_))
            y = t[tuple(sl)]
        elif op == "Div":
            y = t / v[i[1]]
        elif op == "Max":
            y = torch.maximum(t, v[i[1]])
        elif op == "Where":
            y = torch.where(t != 0, v[i[1]], v[i[2]])
        elif op == "ReduceL2":
            y = torch.linalg.vector_norm(t, ord=2, dim=a["axes"], keepdim=bool(a["keepdims"]))
        elif op == "ReduceSum":
            y = t.sum(dim=[int(q) for q in v[i[1]].reshape(-1)], keepdim=bool(a["keepdims"]))
        elif op == "ConvTranspose":
            p = a.get("pads", [0, 0, 0, 0])
            assert p[0] == p[2] and p[1] == p[3]
            y = TF.conv_transpose2d(t, v[i[1]], v[i[2]] if len(i) > 2 else None, stride=a["strides"], padding=(p[0], p[1]), output_padding=a.get("output_padding", [0, 0]))
        else:
            raise NotImplementedError(op)
        v[o] = y
    return v


def swin():
    """the DPT / Swin-class graph: torch evaluates the node list, the numpy oracle must agree, both are stored"""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
    import depth_oracle as DO
    Wt = OX.swin_weights(SEED + 2)
    H = OX.SWIN["H"]
    x = np.random.default_rng(SEED + 3).standard_normal((1, 3, H, H)).astype(np.float32)
    with torch.no_grad():
        v = run_torch(OX.swin_spec(), OX.swin_consts(Wt), x)
    out = v["output"].numpy()
    ref = DO.run_graph(OX.swin_spec(), OX.swin_consts(Wt), {"input": x})
    scale = float(np.abs(out).max())
    assert out.shape == (1, H, H) and float((out > 0).mean()) > 0.2
    for k in ("b1_x2", "b2_a5w", "b2_x2", "xm", "x2g", "u1", "output"):
        err = float(np.abs(v[k].numpy() - ref[k]).max()) / float(np.abs(ref[k]).max())
        assert err < 2e-5, (k, err)
    taps = {k: v[k].numpy() for k in ("b2_x2", "x2g", "u1")}
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "depth_swin.npz"), seed=SEED + 2, input=x, output=out,
                        **{"tap_" + k: a.astype(np.float32) for k, a in taps.items()})
    print("depth_swin.npz: output range", out.min(), out.max(), "positive frac", float((out > 0).mean()), "scale", scale)


def main():
    torch.set_num_threads(1)
    Wt = OX.depth_weights(SEED)
    rng = np.random.default_rng(SEED + 1)
    x = rng.standard_normal((1, 3, H, W)).astype(np.float32)
    with torch.no_grad():
        v = run_torch(OX.depth_spec(), OX.depth_consts(Wt), x)
    out = v["output"].numpy()
    assert out.shape == (1, H, W) and float(out.max()) > float(out.min())
    taps = {k: v[k].numpy() for k in ("b2", "up1")}
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "depth_net.npz"), seed=SEED, input=x, output=out,
                        **{"tap_" + k: a.astype(np.float16) for k, a in taps.items()})
    print("depth_net.npz: output range", out.min(), out.max(), "nonzero frac", float((out > 0).mean()))


if __name__ == "__main__":
    main()
    swin()
