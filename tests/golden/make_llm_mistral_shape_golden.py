"""Pin the LLM oracle against HF transformers at MISTRAL-7B GEOMETRY (2 layers of it): d_model 4096, 32 heads of 128, 8 KV heads (GQA 4 : 1),
d_ff 14336, vocab 32000, the canonical K-split plan (4, 4, 1, 7) — what llm_tiny.npz (head_dim 64, no K-split) cannot show.

Run in the build container only (needs torch + transformers, ~6 GB of RAM):
    python tests/golden/make_llm_mistral_shape_golden.py
Weights: the oracle's synthetic Q4_K_M checkpoint (seed 4, the generator the GPU uses too), de-quantised by the oracle and loaded into
transformers.MistralForCausalLM; HF fp32 prefill of 40 tokens + 6 greedy steps from HF's own KV cache.  Stored: the tokens, HF's logits at
a fixed set of 512 vocabulary columns (+ its arg-max per position), HF's step ids, and the oracle's figures for reference.
tests/test_oracle_llm.py re-runs the oracle against them.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as O  # noqa: E402
from make_llm_golden import unpermute  # noqa: E402

N_PROMPT, N_STEPS, SEED = 40, 6, 4


def main():
    import torch
    from transformers import MistralConfig, MistralForCausalLM

    cfg = O.mistral7b_config(n_layer=2, max_ctx=64, max_seq=1)
    orc = O.OracleLlm(cfg, seed=SEED)
    D, H, KV, HD, FF, V = cfg.d_model, cfg.n_head, cfg.n_kv_head, cfg.head_dim, cfg.d_ff, cfg.vocab
    hf_cfg = MistralConfig(vocab_size=V, hidden_size=D, intermediate_size=FF, num_hidden_layers=cfg.n_layer, num_attention_heads=H,
                           num_key_value_heads=KV, head_dim=HD, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
                           max_position_embeddings=cfg.max_ctx, sliding_window=None, tie_word_embeddings=False, attn_implementation="eager")
    model = MistralForCausalLM(hf_cfg).to(torch.float32).eval()
    sd = {"model.embed_tokens.weight": orc.dequant(-1, O.T_TOKEN_EMBD, V, D), "model.norm.weight": orc.dequant(-1, O.T_OUT_NORM, 1, D)[0],
          "lm_head.weight": orc.dequant(-1, O.T_OUTPUT, V, D)}
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
    del sd

    rng = np.random.default_rng(7)
    tokens = rng.integers(3, V, size=N_PROMPT).astype(np.int32)
    cols = np.sort(rng.choice(V, 512, replace=False)).astype(np.int32)
    with torch.no_grad():
        o = model(torch.from_numpy(tokens.astype(np.int64))[None], use_cache=True)
        hf_logits = o.logits[0].numpy().astype(np.float32)
        past, step_logits, ids = o.past_key_values, [], []
        nxt = int(o.logits[0, -1].argmax())
        for _ in range(N_STEPS):
            ids.append(nxt)
            o = model(torch.tensor([[nxt]]), past_key_values=past, use_cache=True)
            past = o.past_key_values
            step_logits.append(o.logits[0, -1].numpy().astype(np.float32))
            nxt = int(o.logits[0, -1].argmax())
    ids, step_logits = np.array(ids, np.int32), np.stack(step_logits)

    res = {}
    for mode in (1, 0):  # fp32 activations (structure check), then the product's int8 activations
        O.lib().orc_set_fp32_activations(mode)
        orc.reset()
        pl, pam = orc.forward(np.zeros(N_PROMPT, np.int32), np.arange(N_PROMPT, dtype=np.int32), tokens)
        got = np.stack([orc.forward([0], [N_PROMPT + i], [ids[i]])[0][0] for i in range(N_STEPS)])
        res[mode] = (pl, pam, got)
    O.lib().orc_set_fp32_activations(0)
    e32 = max(np.abs(res[1][0] - hf_logits).max(), np.abs(res[1][2] - step_logits).max())
    scale = np.abs(hf_logits).max()
    print(f"fp32-activation mode: max|oracle - hf| = {e32:.3e} (max|logit| {scale:.3f}); arg-max agreement {(res[1][1] == hf_logits.argmax(1)).mean():.3f}")
    assert e32 < 2e-3 * max(1.0, scale)
    assert res[1][1][-1] == ids[0] and np.array_equal(res[1][2].argmax(1)[:-1], ids[1:])
    eq = max(np.abs(res[0][0] - hf_logits).max(), np.abs(res[0][2] - step_logits).max())
    print(f"int8-activation mode: max|oracle - hf| = {eq:.3e}; arg-max agreement {(res[0][1] == hf_logits.argmax(1)).mean():.3f}")
    assert eq < 0.05 * scale
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "llm_mistral_shape.npz")
    np.savez_compressed(out, tokens=tokens, cols=cols, hf_logits_cols=hf_logits[:, cols], hf_argmax=hf_logits.argmax(1).astype(np.int32), hf_ids=ids,
                        hf_step_logits_cols=step_logits[:, cols], hf_scale=np.float32(scale), seed=np.int64(SEED), n_layer=np.int32(cfg.n_layer),
                        fp32_mode_err=np.float32(e32), int8_mode_err=np.float32(eq))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
