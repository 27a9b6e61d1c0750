"""Pin the LLM oracle against an INDEPENDENT fp32 implementation (HF transformers Mistral).

Run in the build container only (needs torch + transformers):
    python tests/golden/make_llm_golden.py
It builds the tiny synthetic checkpoint with the oracle's generator, de-quantises it, loads the
weights into transformers.MistralForCausalLM (q/k rows un-permuted from the GGUF llama-arch
interleaved-pair RoPE convention back to HF's split-half convention), runs HF fp32 prefill, and
stores  tests/golden/llm_tiny.npz  with the token ids, the HF logits and the oracle logits.

The oracle quantises activations to int8 per 256-block (llama.cpp's CPU numerics), HF does
not, so agreement is to quantisation noise (asserted here and re-asserted from the fixture in
tests/test_oracle_llm.py); what this pins is the STRUCTURE: RoPE convention, GQA head mapping,
norm placement, SwiGLU, causal masking, KV-cache indexing, k-quant codecs.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle_lib as O  # noqa: E402


def unpermute(w, n_head):
    # GGUF row 2i / 2i+1 of a head  <-  HF row i / i+half
    rows, cols = w.shape
    hd = rows // n_head
    w = w.reshape(n_head, hd // 2, 2, cols)        # [head][i][pair][col]
    return np.ascontiguousarray(w.transpose(0, 2, 1, 3)).reshape(rows, cols)


def main():
    import torch
    from transformers import MistralConfig, MistralForCausalLM

    cfg = O.tiny_config()
    orc = O.OracleLlm(cfg, seed=4)
    D, H, KV, HD, FF, V = cfg.d_model, cfg.n_head, cfg.n_kv_head, cfg.head_dim, cfg.d_ff, cfg.vocab

    hf_cfg = MistralConfig(vocab_size=V, hidden_size=D, intermediate_size=FF, num_hidden_layers=cfg.n_layer,
                           num_attention_heads=H, num_key_value_heads=KV, head_dim=HD, rms_norm_eps=cfg.rms_eps,
                           rope_theta=cfg.rope_theta, max_position_embeddings=cfg.max_ctx, sliding_window=None,
                           tie_word_embeddings=False, attn_implementation="eager")
    model = MistralForCausalLM(hf_cfg).to(torch.float32).eval()
    sd = {}
    sd["model.embed_tokens.weight"] = orc.dequant(-1, O.T_TOKEN_EMBD, V, D)
    sd["model.norm.weight"] = orc.dequant(-1, O.T_OUT_NORM, 1, D)[0]
    sd["lm_head.weight"] = orc.dequant(-1, O.T_OUTPUT, V, D)
    for l in range(cfg.n_layer):
        p = f"model.layers.{l}."
        sd[p + "input_layernorm.weight"] = orc.dequant(l, O.L_ATTN_NORM, 1, D)[0]
        sd[p + "post_attention_layernorm.weight"] = orc.dequant(l, O.L_FFN_NORM, 1, D)[0]
        sd[p + "self_attn.q_proj.weight"] = unpermute(orc.dequant(l, O.L_Q, H * HD, D), H)
        sd[p + "self_attn.k_proj.weight"] = unpermute(orc.dequant(l, O.L_K, KV * HD, D), KV)
        sd[p + "self_attn.v_proj.weight"] = orc.dequant(l, O.L_V, KV * HD, D)
        sd[p + "self_attn.o_proj.weight"] = orc.dequant(l, O.L_O, D, H * HD)
        sd[p + "mlp.gate_proj.weight"] = orc.dequant(l, O.L_GATE, FF, D)
        sd[p + "mlp.up_proj.weight"] = orc.dequant(l, O.L_UP, FF, D)
        sd[p + "mlp.down_proj.weight"] = orc.dequant(l, O.L_DOWN, D, FF)
    missing = model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=False)
    assert not [k for k in missing.missing_keys if "rotary" not in k], missing

    rng = np.random.default_rng(3)
    tokens = rng.integers(3, V, size=24).astype(np.int32)
    with torch.no_grad():
        hf_logits = model(torch.from_numpy(tokens.astype(np.int64))[None]).logits[0].numpy().astype(np.float32)

    n = len(tokens)
    # structure check: with activation quantisation switched off the oracle must match HF to fp32 rounding
    # (K/V still pass through the f16 cache, hence 2e-3 rather than 1e-5)
    O.lib().orc_set_fp32_activations(1)
    l32, _ = orc.forward(np.zeros(n, np.int32), np.arange(n, dtype=np.int32), tokens)
    O.lib().orc_set_fp32_activations(0)
    orc.reset()
    e32 = np.abs(l32 - hf_logits).max()
    print(f"fp32-activation mode: max|oracle-hf| = {e32:.3e}")
    assert e32 < 2e-3
    logits, am = orc.forward(np.zeros(n, np.int32), np.arange(n, dtype=np.int32), tokens)
    err = np.abs(logits - hf_logits).max()
    scale = np.abs(hf_logits).max()
    agree = (logits.argmax(1) == hf_logits.argmax(1)).mean()
    print(f"max|oracle-hf| = {err:.4e}  (max|logit| = {scale:.3f}), argmax agreement = {agree:.3f}")
    assert err < 0.05 * scale, "oracle disagrees with HF Mistral beyond activation-quantisation noise"
    # ---- decode with the KV cache: HF prefills 16 tokens, then takes 8 greedy single-token steps from its own cache; the oracle is
    # teacher-forced with HF's ids (so a near-tie cannot fork the two sequences) and must reproduce every step's logits: pins cache
    # indexing, position handling and the single-row path against an independent implementation (llm_tiny_decode.npz)
    P, S = 16, 8
    with torch.no_grad():
        o = model(torch.from_numpy(tokens[:P].astype(np.int64))[None], use_cache=True)
        past, step_logits, ids = o.past_key_values, [], []
        nxt = int(o.logits[0, -1].argmax())
        for i in range(S):
            ids.append(nxt)
            o = model(torch.tensor([[nxt]]), past_key_values=past, use_cache=True)
            past = o.past_key_values
            step_logits.append(o.logits[0, -1].numpy().astype(np.float32))
            nxt = int(o.logits[0, -1].argmax())
    ids, step_logits = np.array(ids, np.int32), np.stack(step_logits)
    dec = {}
    for mode in (1, 0):
        O.lib().orc_set_fp32_activations(mode)
        orc.reset()
        pl, pam = orc.forward(np.zeros(P, np.int32), np.arange(P, dtype=np.int32), tokens[:P])
        got = [orc.forward([0], [P + i], [ids[i]])[0][0] for i in range(S)]
        dec[mode] = (pam[-1], np.stack(got))
    O.lib().orc_set_fp32_activations(0)
    e_dec = np.abs(dec[1][1] - step_logits).max()
    print(f"decode, fp32-activation mode: first id {dec[1][0]} vs HF {ids[0]}, max|oracle-hf| over {S} cached steps = {e_dec:.3e}")
    assert dec[1][0] == ids[0] and e_dec < 2e-3
    assert np.array_equal(dec[1][1].argmax(1)[:-1], ids[1:]), "fp32-mode oracle must sample HF's ids step by step"
    eq = np.abs(dec[0][1] - step_logits).max()
    print(f"decode, int8-activation mode: max|oracle-hf| = {eq:.3e}")
    assert eq < 0.05 * np.abs(step_logits).max()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "llm_tiny_decode.npz"), tokens=tokens[:P], hf_ids=ids,
                        hf_step_logits=step_logits, oracle_step_logits=dec[0][1], oracle_first_id=np.int32(dec[0][0]), seed=np.int64(4),
                        fp32_mode_err=np.float32(e_dec))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "llm_tiny.npz")
    np.savez_compressed(out, tokens=tokens, hf_logits=hf_logits.astype(np.float16), oracle_logits=logits,
                        oracle_argmax=am, seed=np.int64(4), fp32_mode_err=np.float32(e32))
    print("wrote", out)


if __name__ == "__main__":
    main()
