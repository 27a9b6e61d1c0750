"""Golden fixtures for the detector stream (run in the build container only).

  preprocess_small.npz   — outputs of the COMPILED reference function
                           (/root/reference/src/vision/tk_image_preprocessor.c via oracle/_ref) on seeded frames,
                           plus sha256 of its output on the BASELINE-size frames (640x480 -> 640x640, 640x640 -> 640x640,
                           and the reference test's gray-128 frame, tests/tk_cortex_test.cpp:79-84).
  yolo_tiny.npz          — YOLOv8n raw head maps on a 64x64 input from an INDEPENDENT torch implementation of the
                           published Ultralytics graph, fed with the oracle's synthetic weights.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402


def frames():
    rng = np.random.default_rng(1)
    return {
        "rand_96x64": rng.integers(0, 256, (64, 96, 3), dtype=np.uint8),
        "rand_37x23": rng.integers(0, 256, (23, 37, 3), dtype=np.uint8),
        "rand_640x480": rng.integers(0, 256, (480, 640, 3), dtype=np.uint8),
        "rand_640x640": rng.integers(0, 256, (640, 640, 3), dtype=np.uint8),
        "gray128_640x480": np.full((480, 640, 3), 128, np.uint8),
    }


def make_preprocess():
    assert O.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    fr = frames()
    out = {}
    out["small_a"] = O.ref_preprocess(fr["rand_96x64"], 64, 64)
    out["small_b"] = O.ref_preprocess(fr["rand_37x23"], 32, 32)
    for name in ("rand_640x480", "rand_640x640", "gray128_640x480"):
        y = O.ref_preprocess(fr[name], 640, 640)
        out["sha_" + name] = np.frombuffer(hashlib.sha256(y.tobytes()).digest(), np.uint8)
        # the restatement must agree with the compiled reference bit for bit
        assert np.array_equal(O.preprocess(fr[name], 640, 640).view(np.uint32), y.view(np.uint32)), name
    assert np.array_equal(O.preprocess(fr["rand_96x64"], 64, 64).view(np.uint32), out["small_a"].view(np.uint32))
    np.savez_compressed(os.path.join(HERE, "preprocess_small.npz"), **out)
    print("wrote preprocess_small.npz (restatement == compiled reference on all five frames)")


def torch_yolov8n(layers, x, nc):
    import torch
    import torch.nn.functional as F
    it = iter(layers)

    def conv(t, act=True):
        L = next(it)
        w = torch.from_numpy(L["w"]).permute(0, 3, 1, 2).contiguous()  # [cout][ky][kx][cin] -> OIHW
        y = F.conv2d(t, w, torch.from_numpy(L["b"]), stride=L["s"], padding=L["k"] // 2)
        assert bool(L["act"]) == act
        return F.silu(y) if act else y

    def c2f(t, n, shortcut):
        y = list(conv(t).chunk(2, 1))
        for _ in range(n):
            z = conv(conv(y[-1]))
            y.append(y[-1] + z if shortcut else z)
        return conv(torch.cat(y, 1))

    def sppf(t):
        t = conv(t)
        y1 = F.max_pool2d(t, 5, 1, 2)
        y2 = F.max_pool2d(y1, 5, 1, 2)
        y3 = F.max_pool2d(y2, 5, 1, 2)
        return conv(torch.cat([t, y1, y2, y3], 1))

    t = conv(conv(x))
    t = c2f(t, 1, True)
    f4 = c2f(conv(t), 2, True)
    f6 = c2f(conv(f4), 2, True)
    t = c2f(conv(f6), 1, True)
    f9 = sppf(t)
    up = lambda v: F.interpolate(v, scale_factor=2, mode="nearest")
    h12 = c2f(torch.cat([up(f9), f6], 1), 1, False)
    h15 = c2f(torch.cat([up(h12), f4], 1), 1, False)
    # NB: the graph header allocates the down-conv before the C2f of the same stage, same order here
    d = conv(h15)
    h18 = c2f(torch.cat([d, h12], 1), 1, False)
    d = conv(h18)
    h21 = c2f(torch.cat([d, f9], 1), 1, False)
    outs = []
    for f in (h15, h18, h21):
        b = conv(conv(conv(f)), act=False)
        c = conv(conv(conv(f)), act=False)
        o = torch.cat([b, c], 1)  # [1, 64+nc, H, W]
        outs.append(o.permute(0, 2, 3, 1).reshape(1, -1, 64 + nc))
    return torch.cat(outs, 1).numpy()


def make_yolo():
    import torch
    torch.manual_seed(0)
    orc = O.OracleYolo(nc=80, seed=5, cls_bias=-4.0)
    rng = np.random.default_rng(7)
    x = rng.standard_normal((1, 64, 64, 3)).astype(np.float32)
    raw = orc.forward(x)
    with torch.no_grad():
        ref = torch_yolov8n(orc.layers(), torch.from_numpy(x).permute(0, 3, 1, 2).contiguous(), 80)
    err = np.abs(raw - ref).max()
    print(f"oracle vs independent torch YOLOv8n: max abs diff {err:.3e} (max |v| {np.abs(ref).max():.3f})")
    assert err < 2e-4 * max(1.0, np.abs(ref).max())
    np.savez_compressed(os.path.join(HERE, "yolo_tiny.npz"), x=x, torch_raw=ref.astype(np.float32), oracle_raw=raw)
    print("wrote yolo_tiny.npz")


def yolo_full_input():
    """the 640 x 640 frame of yolo_full_640.npz: seeded, so the fixture stores no input"""
    return np.random.default_rng(11).standard_normal((1, 640, 640, 3)).astype(np.float32)


def make_yolo_full():
    """yolo_full_640.npz — YOLOv8n at the BASELINE geometry (640 x 640 input, 8400 anchors x 144 channels): the INDEPENDENT torch graph's raw
    head maps, kept as 2048 sampled values + per-scale statistics (a few KB).  Pins the oracle's graph walker (which the oracle shares with the
    product, csrc/common/tk_yolov8n_graph.h) against a second implementation at the size the bench runs, not only at 64 x 64."""
    import torch
    orc = O.OracleYolo(nc=80, seed=5, cls_bias=-4.0)
    x = yolo_full_input()
    with torch.no_grad():
        ref = torch_yolov8n(orc.layers(), torch.from_numpy(x).permute(0, 3, 1, 2).contiguous(), 80)[0]  # [8400][144]
    assert ref.shape == (8400, 144)
    raw = orc.forward(x)[0]
    scale = float(np.abs(ref).max())
    err = float(np.abs(raw - ref).max())
    print(f"oracle vs independent torch YOLOv8n at 640 x 640: max abs diff {err:.3e} (max |v| {scale:.3f})")
    assert err < 2e-4 * max(1.0, scale)
    rng = np.random.default_rng(12)
    idx = np.stack([rng.integers(0, 8400, 2048), rng.integers(0, 144, 2048)], 1).astype(np.int32)
    idx[:96, 0] = np.repeat([0, 79, 80, 6399, 6400, 6401, 7999, 8000, 8001, 8399, 4242, 8123], 8)  # scale seams and corners
    stats = np.array([[ref[a:b].mean(), np.abs(ref[a:b]).max(), ref[a:b].std()] for a, b in ((0, 6400), (6400, 8000), (8000, 8400))], np.float64)
    np.savez_compressed(os.path.join(HERE, "yolo_full_640.npz"), idx=idx, torch_vals=ref[idx[:, 0], idx[:, 1]].astype(np.float32), stats=stats,
                        scale=np.float32(scale))
    print("wrote yolo_full_640.npz")


def torch_yolo_post(raw, H, W, nc, conf, iou, max_det=500, max_cand=2048):
    """INDEPENDENT decode + NMS of a YOLOv8 head in the published Ultralytics / torchvision formulation (vectorised torch): DFL =
    softmax over 16 bins . arange(16); dist2bbox(xyxy) on anchor centres (x + 0.5, y + 0.5) per stride; scores = sigmoid; class =
    arg max; candidates score > conf ordered by score; class-aware NMS by the coordinate-offset trick of torchvision.ops.batched_nms
    (boxes + cls * (max_coord + 1)) with a greedy loop over a vectorised IoU matrix.  Shares no code with csrc/common/tk_yolo_post.h."""
    import torch
    raw = torch.from_numpy(np.ascontiguousarray(raw, np.float32))
    anchors, strides = [], []
    for st in (8, 16, 32):
        h, w = H // st, W // st
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5, torch.arange(w, dtype=torch.float32) + 0.5, indexing="ij")
        anchors.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], 1))
        strides.append(torch.full((h * w,), float(st)))
    anchors, strides = torch.cat(anchors), torch.cat(strides)
    dist = (torch.softmax(raw[:, :64].reshape(-1, 4, 16), -1) * torch.arange(16, dtype=torch.float32)).sum(-1)
    xyxy = torch.cat([anchors - dist[:, :2], anchors + dist[:, 2:]], 1) * strides[:, None]
    scores, cls = torch.sigmoid(raw[:, 64:64 + nc]).max(1)
    idx = torch.nonzero(scores > conf).reshape(-1)
    order = torch.argsort(scores[idx], descending=True, stable=True)   # stable: equal scores keep ascending anchor order
    idx = idx[order][:max_cand]
    b, s, c = xyxy[idx], scores[idx], cls[idx]
    off = b + (c.to(torch.float32) * (b.abs().max() + 1.0) if len(b) else 0.0)[:, None] if len(b) else b
    area = (off[:, 2] - off[:, 0]) * (off[:, 3] - off[:, 1])
    lt = torch.maximum(off[:, None, :2], off[None, :, :2])
    rb = torch.minimum(off[:, None, 2:], off[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    ioum = inter / (area[:, None] + area[None, :] - inter)
    keep, dead = [], torch.zeros(len(b), dtype=torch.bool)
    for i in range(len(b)):
        if dead[i]:
            continue
        keep.append(i)
        dead |= ioum[i] > iou
    keep = torch.tensor(keep[:max_det], dtype=torch.long)
    return (torch.cat([b[keep], s[keep, None]], 1).numpy(), c[keep].numpy().astype(np.int32), idx[keep].numpy().astype(np.int32),
            int(len(b)), ioum.numpy(), c.numpy())


def make_yolo_dets():
    """yolo_tiny_dets.npz — detections of the independent torch post-processor on the oracle's raw head maps (160 x 160 input)."""
    orc = O.OracleYolo(nc=80, seed=5, cls_bias=-1.0)
    rng = np.random.default_rng(19)
    x = rng.standard_normal((1, 160, 160, 3)).astype(np.float32)
    raw = orc.forward(x)[0]
    conf = 0.3
    # the IoU threshold of the fixture is the one (of a few) whose nearest same-class pair is farthest from it: decisions that sit on
    # a rounding boundary would make the comparison of two float implementations a coin toss
    _, _, _, _, ioum, candc = torch_yolo_post(raw, 160, 160, 80, conf, 0.5)
    same = (candc[:, None] == candc[None, :]) & ~np.eye(len(candc), dtype=bool)
    iou, margin = max(((t, float(np.abs(ioum[same] - t).min())) for t in (0.4, 0.45, 0.5, 0.55, 0.6, 0.65)), key=lambda p: p[1])
    tb, tc, ta, ncand, ioum, candc = torch_yolo_post(raw, 160, 160, 80, conf, iou)
    ob, oc, oa = orc.post(raw, 160, 160, conf, iou)
    print(f"iou threshold {iou}: candidates {ncand}, kept by torch {len(ta)}, by the oracle {len(oa)}, closest same-class IoU to the threshold {margin:.2e}")
    assert ncand - len(ta) >= 3, "fixture must exercise suppression"
    assert margin > 2e-5
    assert np.array_equal(ta, oa) and np.array_equal(tc, oc)
    assert np.abs(tb - ob).max() < 1e-3
    np.savez_compressed(os.path.join(HERE, "yolo_tiny_dets.npz"), x=x, conf=np.float32(conf), iou=np.float32(iou), torch_boxes=tb, torch_cls=tc,
                        torch_anchors=ta, n_candidates=np.int32(ncand))
    print("wrote yolo_tiny_dets.npz")


def torch_yolo5(Wt, x, nc):
    """the YOLOv5u-class detector as torch modules in the published Ultralytics definitions (Conv = Conv2d + SiLU, Bottleneck, C3, SPPF, the
    anchor-free Detect head's inference path: make_anchors, DFL, dist2bbox(xywh), sigmoid) — written from those definitions, not from
    tests/onnx_util.yolo5_spec(): it takes the weights by module path.  x NCHW -> [1, 4 + nc, anchors]."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    import onnx_util as OX

    mods = {m[0]: m for m in OX.yolo5_modules(nc)}

    class Conv(nn.Module):
        def __init__(self, name):
            super().__init__()
            _, ci, co, k, s, act = mods[name]
            self.conv = nn.Conv2d(ci, co, k, s, 2 if k == 6 else k // 2)
            self.conv.weight.data = torch.from_numpy(Wt[name + ".w"])
            self.conv.bias.data = torch.from_numpy(Wt[name + ".b"])
            self.act = nn.SiLU() if act else nn.Identity()

        def forward(self, t):
            return self.act(self.conv(t))

    class Bottleneck(nn.Module):
        def __init__(self, name, shortcut):
            super().__init__()
            self.cv1, self.cv2, self.add = Conv(name + ".cv1"), Conv(name + ".cv2"), shortcut

        def forward(self, t):
            return t + self.cv2(self.cv1(t)) if self.add else self.cv2(self.cv1(t))

    class C3(nn.Module):
        def __init__(self, name, n, shortcut=True):
            super().__init__()
            self.cv1, self.cv2, self.cv3 = Conv(name + ".cv1"), Conv(name + ".cv2"), Conv(name + ".cv3")
            self.m = nn.Sequential(*[Bottleneck("%s.m%d" % (name, i), shortcut) for i in range(n)])

        def forward(self, t):
            return self.cv3(torch.cat((self.m(self.cv1(t)), self.cv2(t)), 1))

    class SPPF(nn.Module):
        def __init__(self, name):
            super().__init__()
            self.cv1, self.cv2, self.m = Conv(name + ".cv1"), Conv(name + ".cv2"), nn.MaxPool2d(5, 1, 2)

        def forward(self, t):
            y = [self.cv1(t)]
            y.extend(self.m(y[-1]) for _ in range(3))
            return self.cv2(torch.cat(y, 1))

    class Detect(nn.Module):
        def __init__(self):
            super().__init__()
            self.cv2 = nn.ModuleList(nn.Sequential(*[Conv("d.cv2.%d.%d" % (i, j)) for j in range(3)]) for i in range(3))
            self.cv3 = nn.ModuleList(nn.Sequential(*[Conv("d.cv3.%d.%d" % (i, j)) for j in range(3)]) for i in range(3))

        def forward(self, feats):
            xs = [torch.cat((self.cv2[i](f), self.cv3[i](f)), 1) for i, f in enumerate(feats)]
            pts, strides = [], []
            for f, st in zip(xs, (8.0, 16.0, 32.0)):                    # make_anchors(feats, strides, 0.5)
                h, w = f.shape[2:]
                sy, sx = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5, torch.arange(w, dtype=torch.float32) + 0.5, indexing="ij")
                pts.append(torch.stack((sx, sy), -1).view(-1, 2))
                strides.append(torch.full((h * w, 1), st))
            anchors, strides = torch.cat(pts).transpose(0, 1), torch.cat(strides).transpose(0, 1)
            cat = torch.cat([xi.view(1, 64 + nc, -1) for xi in xs], 2)
            box, cls = cat.split((64, nc), 1)
            b, _, a = box.shape                                          # DFL
            dist = F.conv2d(box.view(b, 4, 16, a).transpose(2, 1).softmax(1), torch.arange(16, dtype=torch.float32).view(1, 16, 1, 1)).view(b, 4, a)
            lt, rb = dist.chunk(2, 1)                                    # dist2bbox(xywh=True, dim=1)
            x1y1, x2y2 = anchors.unsqueeze(0) - lt, anchors.unsqueeze(0) + rb
            dbox = torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1) * strides
            return torch.cat((dbox, cls.sigmoid()), 1)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            D = OX.YOLO5_DEPTH
            self.b0, self.b1, self.b2, self.b3, self.b4 = Conv("b0"), Conv("b1"), C3("b2", D[0]), Conv("b3"), C3("b4", D[1])
            self.b5, self.b6, self.b7, self.b8, self.b9 = Conv("b5"), C3("b6", D[2]), Conv("b7"), C3("b8", D[3]), SPPF("b9")
            self.h10, self.h13, self.h14, self.h17 = Conv("h10"), C3("h13", 1, False), Conv("h14"), C3("h17", 1, False)
            self.h18, self.h20, self.h21, self.h23 = Conv("h18"), C3("h20", 1, False), Conv("h21"), C3("h23", 1, False)
            self.up, self.detect = nn.Upsample(None, 2, "nearest"), Detect()

        def forward(self, t):
            p3 = self.b4(self.b3(self.b2(self.b1(self.b0(t)))))
            p4 = self.b6(self.b5(p3))
            x9 = self.b9(self.b8(self.b7(p4)))
            h10 = self.h10(x9)
            h14 = self.h14(self.h13(torch.cat((self.up(h10), p4), 1)))
            h17 = self.h17(torch.cat((self.up(h14), p3), 1))
            h20 = self.h20(torch.cat((self.h18(h17), h14), 1))
            h23 = self.h23(torch.cat((self.h21(h20), h10), 1))
            return self.detect([h17, h20, h23])

    with torch.no_grad():
        return Net().eval()(torch.from_numpy(x)).numpy()


def torch_post_decoded(out, nc, conf, iou, max_det=500, max_cand=2048):
    """decode + class-aware NMS of an Ultralytics export's [4 + nc][anchors] output (xywh centre boxes in input pixels, probabilities): xyxy,
    score = max over classes, candidates score > conf by (score desc, anchor asc), torchvision's batched-NMS coordinate-offset trick"""
    import torch
    o = torch.from_numpy(np.ascontiguousarray(out, np.float32))
    cx, cy, w, h = o[0], o[1], o[2], o[3]
    xyxy = torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    scores, cls = o[4:4 + nc].max(0)
    idx = torch.nonzero(scores > conf).reshape(-1)
    idx = idx[torch.argsort(scores[idx], descending=True, stable=True)][:max_cand]
    b, s, c = xyxy[idx], scores[idx], cls[idx]
    off = b + (c.to(torch.float32) * (b.abs().max() + 1.0))[:, None] if len(b) else b
    area = (off[:, 2] - off[:, 0]) * (off[:, 3] - off[:, 1])
    wh = (torch.minimum(off[:, None, 2:], off[None, :, 2:]) - torch.maximum(off[:, None, :2], off[None, :, :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    ioum = inter / (area[:, None] + area[None, :] - inter)
    keep, dead = [], torch.zeros(len(b), dtype=torch.bool)
    for i in range(len(b)):
        if dead[i]:
            continue
        keep.append(i)
        dead |= ioum[i] > iou
    keep = torch.tensor(keep[:max_det], dtype=torch.long)
    return torch.cat([b[keep], s[keep, None]], 1).numpy(), c[keep].numpy().astype(np.int32), idx[keep].numpy().astype(np.int32), int(len(b)), ioum.numpy(), c.numpy()


def make_yolo5():
    """yolo5nu_tiny.npz — the YOLOv5u-class graph (tests/onnx_util.yolo5_model: what the detector loads through the ONNX executor) evaluated by
    the torch modules above on seeded weights and a seeded 128 x 128 input: the [1, 4 + nc, anchors] output and the detections of the
    independent post-processor.  The IoU threshold is picked away from every same-class pair's IoU, as for yolo_tiny_dets.npz."""
    import onnx_util as OX
    nc, H, seed = 80, 128, 21
    Wt = OX.yolo5_weights(seed, nc, cls_bias=-1.2)
    rng = np.random.default_rng(23)
    x = rng.standard_normal((1, 3, H, H)).astype(np.float32)
    out = torch_yolo5(Wt, x, nc)[0]
    conf = 0.3
    _, _, _, _, ioum, candc = torch_post_decoded(out, nc, conf, 0.5)
    same = (candc[:, None] == candc[None, :]) & ~np.eye(len(candc), dtype=bool)
    iou, margin = max(((t, float(np.abs(ioum[same] - t).min()) if same.any() else 1.0) for t in (0.4, 0.45, 0.5, 0.55, 0.6, 0.65)), key=lambda p: p[1])
    tb, tc, ta, ncand, _, _ = torch_post_decoded(out, nc, conf, iou)
    print(f"yolo5: output {out.shape}, scale {np.abs(out).max():.3f}; iou {iou}: candidates {ncand}, kept {len(ta)}, closest same-class IoU to the threshold {margin:.2e}")
    assert ncand >= 8 and ncand - len(ta) >= 2 and margin > 2e-4
    np.savez_compressed(os.path.join(HERE, "yolo5nu_tiny.npz"), x=x, seed=np.int32(seed), cls_bias=np.float32(-1.2), out=out, conf=np.float32(conf), iou=np.float32(iou),
                        torch_boxes=tb, torch_cls=tc, torch_anchors=ta, n_candidates=np.int32(ncand))
    print("wrote yolo5nu_tiny.npz")


def attribute_vector_frames():
    """the two frames of the reference's tests/tk_attribute_classifier_test.c:21-54 (pure red) and :56-91 (gray with two black column bands)"""
    red = np.zeros((100, 100, 3), np.uint8)
    red[..., 0] = 255
    door = np.full((100, 100, 3), 200, np.uint8)
    door[:, 20:25] = 0
    door[:, 70:75] = 0
    return {"red_frame": (red, (10, 10, 80, 80)), "door_frame": (door, (0, 0, 100, 100))}


def make_attribute_vectors():
    """checks attribute_vectors.json against the compiled reference classifiers (the json is hand-written data + these outputs)"""
    import json
    assert O.have_ref()
    j = json.load(open(os.path.join(HERE, "attribute_vectors.json")))
    for name, (frame, box) in attribute_vector_frames().items():
        color, door = O.ref_attributes(frame, box)
        assert j[name]["compiled_reference"] == {"color": color, "door": door}, (name, color, door)
        print(name, "->", color, door, "(reference test expects", j[name]["reference_test_expects"], ")")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "yolo5":
        make_yolo5()
        sys.exit(0)
    make_attribute_vectors()
    make_preprocess()
    make_yolo()
    make_yolo_full()
    make_yolo_dets()
    make_yolo5()
