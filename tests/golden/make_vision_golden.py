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


if __name__ == "__main__":
    make_preprocess()
    make_yolo()
