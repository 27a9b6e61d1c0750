"""vad_graph.npz — speech probabilities of a generated Silero-class VAD graph (tests/onnx_util.py: vad_model, seeded weights) computed by
an independent torch implementation: F.pad(reflect) -> F.conv1d STFT -> magnitude -> conv / ReLU encoder -> an LSTM cell written out in
ONNX's gate order (i, o, f, c) -> 1x1 conv -> sigmoid -> mean; eight consecutive 30 ms windows with the recurrent state carried, then
the same eight from a cleared state after a reset.  No silero_vad.onnx exists offline: this pins the GPU graph executor
(csrc/audio/tk_vad_graph.hip), not the published checkpoint.  Run: python tests/golden/make_vad_graph_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import onnx_util  # noqa: E402


def torch_vad(W, windows, hop=32, hidden=32):
    import torch
    import torch.nn.functional as F
    T = {k: torch.from_numpy(v) for k, v in W.items()}
    n_bins = W["stft"].shape[0] // 2
    n_fft = W["stft"].shape[2]
    h = torch.zeros(hidden)
    c = torch.zeros(hidden)
    out = []
    with torch.no_grad():
        for w in windows:
            x = torch.from_numpy(w)[None, None, :]
            x = F.pad(x, (n_fft // 2, n_fft // 2), mode="reflect")
            spec = F.conv1d(x, T["stft"], stride=hop)
            mag = torch.sqrt(spec[:, :n_bins] ** 2 + spec[:, n_bins:] ** 2)
            e = F.relu(F.conv1d(mag, T["enc1.w"], T["enc1.b"], padding=1))
            e = F.relu(F.conv1d(e, T["enc2.w"], T["enc2.b"], stride=2, padding=1))
            e = F.relu(F.conv1d(e, T["enc3.w"], T["enc3.b"], stride=2, padding=1))
            ys = []
            Wm, Rm, B = T["lstm.W"][0], T["lstm.R"][0], T["lstm.B"][0]
            for t in range(e.shape[2]):
                g = Wm @ e[0, :, t] + Rm @ h + B[:4 * hidden] + B[4 * hidden:]
                i, o, f, cc = torch.sigmoid(g[:hidden]), torch.sigmoid(g[hidden:2 * hidden]), torch.sigmoid(g[2 * hidden:3 * hidden]), torch.tanh(g[3 * hidden:])
                c = f * c + i * cc
                h = o * torch.tanh(c)
                ys.append(h)
            y = torch.stack(ys, 1)[None]                                     # [1, hidden, T]
            p = torch.sigmoid(F.conv1d(F.relu(y), T["head.w"], T["head.b"]))
            out.append(float(p.mean()))
    return np.array(out, np.float32)


def main():
    seed = 21
    W = onnx_util.vad_weights(seed)
    rng = np.random.default_rng(5)
    t = np.arange(8 * 480) / 16000.0
    sig = (0.3 * np.sin(2 * np.pi * 220 * t) * (t > 0.06) + 0.02 * rng.standard_normal(t.size)).astype(np.float32)
    windows = sig.reshape(8, 480)
    probs = torch_vad(W, windows)
    print("torch probabilities:", probs)
    assert probs.std() > 1e-3, "the fixture should not be a constant"
    np.savez_compressed(os.path.join(HERE, "vad_graph.npz"), seed=np.int64(seed), windows=windows, torch_probs=probs)
    print("wrote vad_graph.npz")


if __name__ == "__main__":
    main()
