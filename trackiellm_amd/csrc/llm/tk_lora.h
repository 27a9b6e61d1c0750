/*
 * tk_lora.h — LoRA adapter files, merged into the model's weights at load.
 *
 * The reference applies an adapter right after the model is loaded, in place and once: llama_model_apply_lora_from_file(model, path, ...)
 * (src/ai_models/tk_model_loader.c:259-270; llama.cpp itself is an un-fetched submodule).  That call computes W' = W + (alpha / r) B A per adapted
 * matrix and — for a quantised W without a base model — dequantises, adds and quantises back to W's own type (ggml's add on a quantised
 * destination).  The same happens here on the GPU while a tensor is installed (k_lora_merge, csrc/llm/tk_llm_kernels.hip): nothing changes on
 * the decode path, an adapted model streams exactly the bytes an un-adapted one does.
 *
 * Two containers are read:
 *   - "ggla" (what llama_model_apply_lora_from_file read; convert-lora-to-ggml.py's output): u32 magic 0x67676c61, u32 version 1, i32 r, i32 alpha,
 *     then per tensor { i32 n_dims, i32 name_len, i32 ftype (0 f32, 1 f16) }, n_dims x i32 ne, the name ("blk.3.attn_q.weight.loraA" / ".loraB"),
 *     padding to a multiple of 32 bytes, data.  loraA has ne = {r, k_in} (memory [k_in][r]), loraB ne = {r, n_out} (memory [n_out][r]).
 *   - GGUF adapters (convert_lora_to_gguf.py): general.type = "adapter", adapter.type = "lora", adapter.lora.alpha; tensors
 *     "<base>.lora_a" ne = {k_in, r} (memory [r][k_in]) and "<base>.lora_b" ne = {r, n_out} (memory [n_out][r]), F32 or F16.
 * Both formats restated from llama.cpp's published converters / loaders (third-party, un-vendored: parity unpinned).
 */
#ifndef TK_LORA_H
#define TK_LORA_H

#include <stdint.h>

#include <string>
#include <vector>

struct TkLoraTensor {
    int layer = -1, which = -1;  /* TkLlmLayerTensorId, or layer -1 + TkLlmTensorId for output.weight */
    int64_t n_out = 0, k_in = 0;
    int r = 0;
    std::vector<float> A; /* [r][k_in] */
    std::vector<float> B; /* [n_out][r] */
};

struct TkLoraAdapter {
    std::string error, path;
    int r = 0;          /* rank of the first tensor (GGUF adapters may mix ranks; the scale is alpha / that tensor's r) */
    float alpha = 0.0f;
    std::vector<TkLoraTensor> tensors;

    bool load(const char* path);
    const TkLoraTensor* find(int layer, int which) const;
    /* alpha / r of THAT tensor; an adapter written with alpha = 0 scales by 1 (llama.cpp's llama_lora_adapter: `alpha ? alpha / rank : 1`) */
    float scale_of(const TkLoraTensor& t) const { return alpha != 0.0f ? alpha / (float)t.r : 1.0f; }
    /* a loaded model has the adapter inside its weights: the factors (tens to hundreds of MB of host memory for a 7B model) can go */
    void drop_factors() {
        for (auto& t : tensors) { std::vector<float>().swap(t.A); std::vector<float>().swap(t.B); }
    }
};

#endif
