"""Golden fixtures for the ASR / VAD streams (run in the build container only; needs torch + transformers).

  whisper_tiny.npz  — a small Whisper geometry (2+2 layers, d = 64, 1 s window) with the oracle's synthetic weights loaded
                      into HF transformers' WhisperForConditionalGeneration (activation gelu_new = the tanh GELU ggml
                      evaluates) and HF's WhisperFeatureExtractor: log-mel, encoder states and first-step logits from
                      the INDEPENDENT implementation next to the oracle's, plus the oracle's forced greedy token ids.
  vad_state_machine.json — probability traces -> event sequences of the reference state machine
                      (src/sensors/tk_vad_silero.c:283-322), produced by the oracle's restatement and hand-checkable.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402


def pcm_fixture(n=16000, B=2):
    rng = np.random.default_rng(2)
    t = np.arange(n) / 16000.0
    a = 6000 * np.sin(2 * np.pi * 440 * t) + 3000 * np.sin(2 * np.pi * 1250 * t + 0.3) + rng.normal(0, 500, n)
    b = np.clip(rng.normal(0, 3000, n), -32768, 32767)
    return np.stack([a, b])[:B].astype(np.int16)


def hf_whisper_reference(hp, orc, pcm, mel, chunk_length):
    """HF transformers' feature extractor and WhisperForConditionalGeneration (eager attention, gelu_new = the tanh GELU ggml evaluates) carrying
    the oracle's synthetic weights: (log-mel [B][frames][mels], encoder states, first-step logits) of the INDEPENDENT implementation"""
    import torch
    from transformers import WhisperConfig, WhisperFeatureExtractor, WhisperForConditionalGeneration

    fe = WhisperFeatureExtractor(feature_size=hp.n_mels, sampling_rate=16000, hop_length=160, chunk_length=chunk_length, n_fft=400)
    feats = fe([p.astype(np.float32) / 32768.0 for p in pcm], sampling_rate=16000, return_tensors="np")["input_features"]  # [B][mels][frames]
    hf_mel = np.transpose(feats, (0, 2, 1))

    cfg = WhisperConfig(vocab_size=hp.n_vocab, num_mel_bins=hp.n_mels, encoder_layers=hp.n_audio_layer, encoder_attention_heads=hp.n_audio_head,
                        decoder_layers=hp.n_text_layer, decoder_attention_heads=hp.n_text_head, d_model=hp.n_audio_state,
                        encoder_ffn_dim=4 * hp.n_audio_state, decoder_ffn_dim=4 * hp.n_text_state, max_source_positions=hp.n_audio_ctx,
                        max_target_positions=hp.n_text_ctx, activation_function="gelu_new", dropout=0.0, attention_dropout=0.0,
                        activation_dropout=0.0, scale_embedding=False, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                        decoder_start_token_id=1, attn_implementation="eager")
    model = WhisperForConditionalGeneration(cfg).float().eval()
    T = orc.tensors()
    sd = {}

    def conv(w, cin):  # [d][(kx, c)] -> [d][c][kx]
        return np.ascontiguousarray(w.reshape(w.shape[0], 3, cin).transpose(0, 2, 1))

    sd["model.encoder.conv1.weight"] = conv(T["encoder.conv1.weight"], hp.n_mels)
    sd["model.encoder.conv1.bias"] = T["encoder.conv1.bias"][0]
    sd["model.encoder.conv2.weight"] = conv(T["encoder.conv2.weight"], hp.n_audio_state)
    sd["model.encoder.conv2.bias"] = T["encoder.conv2.bias"][0]
    sd["model.encoder.embed_positions.weight"] = T["encoder.positional_embedding"]
    sd["model.encoder.layer_norm.weight"] = T["encoder.ln_post.weight"][0]
    sd["model.encoder.layer_norm.bias"] = T["encoder.ln_post.bias"][0]
    sd["model.decoder.embed_tokens.weight"] = T["decoder.token_embedding.weight"]
    sd["proj_out.weight"] = T["decoder.token_embedding.weight"]
    sd["model.decoder.embed_positions.weight"] = T["decoder.positional_embedding"]
    sd["model.decoder.layer_norm.weight"] = T["decoder.ln.weight"][0]
    sd["model.decoder.layer_norm.bias"] = T["decoder.ln.bias"][0]

    def block(src, dst, cross):
        m = {"attn_ln": "self_attn_layer_norm", "mlp_ln": "final_layer_norm", "cross_attn_ln": "encoder_attn_layer_norm"}
        for a, b in m.items():
            if a == "cross_attn_ln" and not cross:
                continue
            sd[dst + b + ".weight"] = T[src + a + ".weight"][0]
            sd[dst + b + ".bias"] = T[src + a + ".bias"][0]
        for a, b in (("attn", "self_attn"),) + ((("cross_attn", "encoder_attn"),) if cross else ()):
            for p, q in (("query", "q_proj"), ("key", "k_proj"), ("value", "v_proj"), ("out", "out_proj")):
                sd[dst + b + "." + q + ".weight"] = T[src + a + "." + p + ".weight"]
                if p != "key":
                    sd[dst + b + "." + q + ".bias"] = T[src + a + "." + p + ".bias"][0]
        sd[dst + "fc1.weight"] = T[src + "mlp.0.weight"]; sd[dst + "fc1.bias"] = T[src + "mlp.0.bias"][0]
        sd[dst + "fc2.weight"] = T[src + "mlp.2.weight"]; sd[dst + "fc2.bias"] = T[src + "mlp.2.bias"][0]

    for l in range(hp.n_audio_layer):
        block(f"encoder.blocks.{l}.", f"model.encoder.layers.{l}.", False)
    for l in range(hp.n_text_layer):
        block(f"decoder.blocks.{l}.", f"model.decoder.layers.{l}.", True)
    res = model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=False)
    assert not res.missing_keys and not res.unexpected_keys, res

    prompt = O.whisper_prompt(hp)
    with torch.no_grad():
        out = model(input_features=torch.from_numpy(np.ascontiguousarray(np.transpose(mel, (0, 2, 1)))),
                    decoder_input_ids=torch.from_numpy(np.tile(prompt.astype(np.int64), (len(pcm), 1))), output_hidden_states=False)
        hf_enc = out.encoder_last_hidden_state.numpy()
        hf_lg = out.logits[:, -1].numpy()
    return hf_mel, hf_enc, hf_lg


def make_whisper():
    hp = O.whisper_tiny_test()
    orc = O.OracleWhisper(hp, seed=6)
    pcm = pcm_fixture()
    toks, mel, enc, lg = orc.transcribe(pcm, 6)
    hf_mel, hf_enc, hf_lg = hf_whisper_reference(hp, orc, pcm, mel, 1)
    e_mel = np.abs(hf_mel - mel).max()
    print(f"log-mel: oracle vs HF feature extractor max abs diff {e_mel:.3e}")
    assert e_mel < 2e-4
    e_enc = np.abs(hf_enc - enc).max()
    e_lg = np.abs(hf_lg - lg).max()
    print(f"encoder states: max abs diff {e_enc:.3e} (max |v| {np.abs(hf_enc).max():.2f}); first-step logits: {e_lg:.3e} (max |v| {np.abs(hf_lg).max():.2f})")
    assert e_enc < 2e-4 * max(1.0, np.abs(hf_enc).max()) and e_lg < 2e-4 * max(1.0, np.abs(hf_lg).max())
    assert np.array_equal(hf_lg.argmax(1), toks[:, 0])
    np.savez_compressed(os.path.join(HERE, "whisper_tiny.npz"), pcm=pcm, hf_mel=hf_mel.astype(np.float32), hf_enc=hf_enc.astype(np.float32),
                        hf_logits=hf_lg.astype(np.float32), oracle_tokens=toks, oracle_mel=mel, oracle_logits=lg)
    print("wrote whisper_tiny.npz; oracle tokens", toks.tolist())


def whisper_full_pcm():
    """the 1 s utterance of whisper_full_tiny_en.npz (seeded: the fixture stores no input)"""
    return pcm_fixture(16000, 1)


def make_whisper_full():
    """whisper_full_tiny_en.npz — the tiny.en GEOMETRY of the bench (80 mels, 30 s window -> 1500 positions, d 384, 6 heads, 4 + 4 layers,
    vocabulary 51864): HF's log-mel, encoder states and first-step logits for one 1 s utterance, kept as sampled values + statistics (KB-sized).
    Pins the oracle's graph walker (shared with the product: csrc/common/tk_whisper_graph.h) by an independent implementation at the real size."""
    hp = O.whisper_tiny_en()
    orc = O.OracleWhisper(hp, seed=6)
    pcm = whisper_full_pcm()
    toks, mel, enc, lg = orc.transcribe(pcm, 1)
    hf_mel, hf_enc, hf_lg = hf_whisper_reference(hp, orc, pcm, mel, 30)
    assert hf_mel.shape == mel.shape == (1, 3000, 80) and hf_enc.shape == enc.shape == (1, 1500, 384) and hf_lg.shape == lg.shape == (1, 51864)
    e_mel, e_enc, e_lg = np.abs(hf_mel - mel).max(), np.abs(hf_enc - enc).max(), np.abs(hf_lg - lg).max()
    s_enc, s_lg = float(np.abs(hf_enc).max()), float(np.abs(hf_lg).max())
    print(f"tiny.en geometry: log-mel {e_mel:.3e}; encoder states {e_enc:.3e} (max |v| {s_enc:.2f}); first-step logits {e_lg:.3e} (max |v| {s_lg:.2f})")
    assert e_mel < 2e-4 and e_enc < 2e-4 * max(1.0, s_enc) and e_lg < 2e-4 * max(1.0, s_lg)
    assert int(hf_lg.argmax(1)[0]) == int(toks[0, 0])
    rng = np.random.default_rng(13)
    i_mel = np.stack([rng.integers(0, 3000, 1024), rng.integers(0, 80, 1024)], 1).astype(np.int32)
    i_mel[:256, 0] = rng.integers(0, 110, 256)  # the frames that hold the utterance
    i_enc = np.stack([rng.integers(0, 1500, 2048), rng.integers(0, 384, 2048)], 1).astype(np.int32)
    i_enc[:512, 0] = rng.integers(0, 60, 512)
    i_lg = rng.integers(0, 51864, 2048).astype(np.int32)
    top = np.argsort(-hf_lg[0])[:16].astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "whisper_full_tiny_en.npz"), i_mel=i_mel, hf_mel=hf_mel[0][i_mel[:, 0], i_mel[:, 1]].astype(np.float32),
                        i_enc=i_enc, hf_enc=hf_enc[0][i_enc[:, 0], i_enc[:, 1]].astype(np.float32), enc_scale=np.float32(s_enc),
                        enc_stats=np.array([hf_enc.mean(), hf_enc.std()], np.float64), i_lg=i_lg, hf_lg=hf_lg[0][i_lg].astype(np.float32),
                        lg_scale=np.float32(s_lg), top_ids=top, top_logits=hf_lg[0][top].astype(np.float32))
    print("wrote whisper_full_tiny_en.npz; first token", int(toks[0, 0]))


def make_vad():
    traces = {
        "short_blip_is_ignored": [0.9] * 5 + [0.1] * 20,                       # 150 ms of speech < 250 ms
        "start_then_end": [0.1] * 3 + [0.9] * 12 + [0.2] * 12 + [0.1] * 3,     # 360 ms speech, 360 ms silence
        "silence_gap_shorter_than_min": [0.9] * 10 + [0.1] * 5 + [0.9] * 10 + [0.1] * 11,
        "threshold_edge_is_speech": [0.5] * 9 + [0.49999] * 10,
    }
    out = {}
    for name, tr in traces.items():
        ev, _ = O.vad_run(tr)
        out[name] = {"probabilities": tr, "events": ev}
    # hand check of the simplest one: 9th speech window (index 3+8) reaches 270 ms >= 250 ms; 10 silent windows = 300 ms
    assert out["start_then_end"]["events"] == [(11, 0), (24, 1)], out["start_then_end"]["events"]
    assert out["short_blip_is_ignored"]["events"] == []
    json.dump(out, open(os.path.join(HERE, "vad_state_machine.json"), "w"), indent=1)
    print("wrote vad_state_machine.json")


if __name__ == "__main__":
    make_vad()
    make_whisper()
    make_whisper_full()
