"""Writer of whisper.cpp "ggml" checkpoint files for tests (the layout of whisper.cpp's convert-pt-to-ggml.py: header, mel filter bank,
byte-level vocabulary, then tensors with OpenAI's names; conv kernels [out][in][3], biases [n][1], 2-D weights f16 when ftype = 1)."""
import struct

import numpy as np


def f16_round(a):
    return a.astype(np.float16).astype(np.float32)


def checkpoint_tensors(tensors, hp, f16=True):
    """this path's tensors (oracle_lib.OracleWhisper.tensors()) -> (the same set with 2-D weights rounded to f16, file tensors in
    PyTorch layout).  Frontend tables are not part of a checkpoint (the filter bank travels in the header)."""
    rounded, file_t = {}, []
    for name, a in tensors.items():
        if name.startswith("frontend."):
            continue
        two_d = a.shape[0] > 1 and not name.endswith("positional_embedding")
        w = f16_round(a) if (f16 and two_d) else a
        rounded[name] = w
        if name in ("encoder.conv1.weight", "encoder.conv2.weight"):
            cin = hp.n_mels if "conv1" in name else hp.n_audio_state
            pt = np.ascontiguousarray(w.reshape(w.shape[0], 3, cin).transpose(0, 2, 1))  # [out][(tap, in)] -> [out][in][tap]
        elif name.endswith(".bias") and "conv" in name:
            pt = w.reshape(-1, 1)
        elif a.shape[0] == 1:
            pt = w.reshape(-1)
        else:
            pt = w
        file_t.append((name, pt, 1 if (f16 and two_d) else 0))
    return rounded, file_t


def write_ggml(path, hp, mel_filters, vocab, file_tensors, ftype=1, magic=0x67676D6C):
    with open(path, "wb") as f:
        f.write(struct.pack("<I", magic))
        f.write(struct.pack("<11i", hp.n_vocab, hp.n_audio_ctx, hp.n_audio_state, hp.n_audio_head, hp.n_audio_layer, hp.n_text_ctx,
                            hp.n_text_state, hp.n_text_head, hp.n_text_layer, hp.n_mels, ftype))
        mf = np.ascontiguousarray(mel_filters, np.float32)
        f.write(struct.pack("<2i", mf.shape[0], mf.shape[1]))
        f.write(mf.tobytes())
        f.write(struct.pack("<i", len(vocab)))
        for tok in vocab:
            f.write(struct.pack("<I", len(tok)))
            f.write(tok)
        for name, a, ttype in file_tensors:
            nb = name.encode()
            f.write(struct.pack("<3i", a.ndim, len(nb), ttype))
            for d in reversed(a.shape):
                f.write(struct.pack("<i", d))
            f.write(nb)
            f.write(np.ascontiguousarray(a, np.float16 if ttype == 1 else np.float32).tobytes())
