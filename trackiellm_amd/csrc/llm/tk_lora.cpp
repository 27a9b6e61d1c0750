#include "tk_lora.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>

#include "../common/tk_exact_math.h"
#include "tk_gguf.h"
#include "tk_llm_engine.h"

namespace {

/* "blk.7.ffn_down.weight" -> (7, TK_L_DOWN); "output.weight" -> (-1, TK_T_OUTPUT); norms and the embedding are not adapted */
bool tensor_of(const std::string& base, int* layer, int* which) {
    if (base == "output.weight") { *layer = -1; *which = TK_T_OUTPUT; return true; }
    static const struct { const char* name; int which; } names[] = {
        {"attn_q", TK_L_Q}, {"attn_k", TK_L_K}, {"attn_v", TK_L_V}, {"attn_output", TK_L_O}, {"ffn_gate", TK_L_GATE}, {"ffn_up", TK_L_UP}, {"ffn_down", TK_L_DOWN}};
    int l = -1, used = 0;
    if (sscanf(base.c_str(), "blk.%d.%n", &l, &used) != 1 || l < 0 || used <= 0) return false;
    const std::string rest = base.substr((size_t)used);
    for (const auto& n : names)
        if (rest == std::string(n.name) + ".weight") { *layer = l; *which = n.which; return true; }
    return false;
}

void to_f32(const uint8_t* src, int f16, size_t n, float* dst) {
    if (!f16) { memcpy(dst, src, n * 4); return; }
    for (size_t i = 0; i < n; ++i) {
        uint16_t h;
        memcpy(&h, src + 2 * i, 2);
        dst[i] = tk_f16_to_f32(h);
    }
}

struct Half { /* one of the two factors as it sits in the file */
    const uint8_t* data = nullptr;
    int f16 = 0;
    int64_t ne0 = 0, ne1 = 0;
};

}  // namespace

const TkLoraTensor* TkLoraAdapter::find(int layer, int which) const {
    for (const auto& t : tensors)
        if (t.layer == layer && t.which == which) return &t;
    return nullptr;
}

static bool pair_up(TkLoraAdapter* ad, const std::map<std::string, Half>& as, const std::map<std::string, Half>& bs, bool a_is_transposed) {
    for (const auto& kv : as) {
        auto ib = bs.find(kv.first);
        if (ib == bs.end()) { ad->error = "adapter has " + kv.first + " factor A without factor B"; return false; }
        TkLoraTensor t;
        if (!tensor_of(kv.first, &t.layer, &t.which)) { ad->error = "adapter tensor for an unknown base tensor: " + kv.first; return false; }
        const Half& a = kv.second;
        const Half& b = ib->second;
        /* ggla: A ne = {r, k_in}; GGUF: A ne = {k_in, r}; B ne = {r, n_out} in both */
        t.r = (int)(a_is_transposed ? a.ne0 : a.ne1);
        t.k_in = a_is_transposed ? a.ne1 : a.ne0;
        t.n_out = b.ne1;
        if (t.r < 1 || t.r > 1024 || b.ne0 != t.r || t.k_in < 1 || t.n_out < 1) { ad->error = "adapter factors of " + kv.first + " do not share a rank"; return false; }
        t.A.resize((size_t)t.r * t.k_in);
        t.B.resize((size_t)t.n_out * t.r);
        if (a_is_transposed) { /* file holds [k_in][r] */
            std::vector<float> tmp(t.A.size());
            to_f32(a.data, a.f16, tmp.size(), tmp.data());
            for (int64_t k = 0; k < t.k_in; ++k)
                for (int j = 0; j < t.r; ++j) t.A[(size_t)j * t.k_in + k] = tmp[(size_t)k * t.r + j];
        } else {
            to_f32(a.data, a.f16, t.A.size(), t.A.data());
        }
        to_f32(b.data, b.f16, t.B.size(), t.B.data());
        if (ad->tensors.empty()) ad->r = t.r;
        ad->tensors.push_back(std::move(t));
    }
    for (const auto& kv : bs)
        if (!as.count(kv.first)) { ad->error = "adapter has " + kv.first + " factor B without factor A"; return false; }
    if (ad->tensors.empty()) { ad->error = "adapter holds no tensors"; return false; }
    return true;
}

static bool load_ggla(TkLoraAdapter* ad, const std::vector<uint8_t>& buf) {
    size_t p = 8;
    auto rd = [&](int32_t* v) { if (p + 4 > buf.size()) return false; memcpy(v, buf.data() + p, 4); p += 4; return true; };
    int32_t r = 0, alpha = 0;
    if (!rd(&r) || !rd(&alpha) || r < 1) { ad->error = "corrupt ggla header"; return false; }
    ad->alpha = (float)alpha;
    std::map<std::string, Half> as, bs;
    while (p < buf.size()) {
        int32_t n_dims = 0, name_len = 0, ftype = 0;
        if (!rd(&n_dims) || !rd(&name_len) || !rd(&ftype)) { ad->error = "truncated ggla tensor header"; return false; }
        if (n_dims < 1 || n_dims > 2 || name_len < 1 || name_len > 256 || (ftype != 0 && ftype != 1)) { ad->error = "unsupported ggla tensor (want 1-2 dims, f32 or f16)"; return false; }
        int32_t ne[2] = {1, 1};
        for (int d = 0; d < n_dims; ++d)
            if (!rd(&ne[d]) || ne[d] < 1) { ad->error = "corrupt ggla tensor shape"; return false; }
        if (p + (size_t)name_len > buf.size()) { ad->error = "truncated ggla tensor name"; return false; }
        const std::string name((const char*)buf.data() + p, (size_t)name_len);
        p += (size_t)name_len;
        p = (p + 31) & ~(size_t)31;
        const size_t nbytes = (size_t)ne[0] * (size_t)ne[1] * (ftype ? 2 : 4);
        if (p > buf.size() || nbytes > buf.size() - p) { ad->error = "ggla tensor data runs past the end of the file: " + name; return false; }
        const size_t dot = name.rfind('.');
        const std::string kind = dot == std::string::npos ? std::string() : name.substr(dot + 1);
        if (kind != "loraA" && kind != "loraB") { ad->error = "ggla tensor is neither .loraA nor .loraB: " + name; return false; }
        Half h;
        h.data = buf.data() + p; h.f16 = ftype; h.ne0 = ne[0]; h.ne1 = ne[1];
        (kind == "loraA" ? as : bs)[name.substr(0, dot)] = h;
        p += nbytes;
    }
    return pair_up(ad, as, bs, /*a_is_transposed=*/true);
}

static bool load_gguf(TkLoraAdapter* ad, const char* path) {
    TkGgufFile f;
    if (!f.open(path)) { ad->error = f.error; return false; }
    auto ty = f.str.find("general.type");
    auto at = f.str.find("adapter.type");
    if (ty == f.str.end() || ty->second != "adapter" || at == f.str.end() || at->second != "lora") { ad->error = "GGUF file is not a LoRA adapter (general.type / adapter.type)"; return false; }
    if (!f.num.count("adapter.lora.alpha")) { ad->error = "GGUF adapter lacks adapter.lora.alpha"; return false; }
    ad->alpha = (float)f.get("adapter.lora.alpha", 0.0);
    std::map<std::string, Half> as, bs;
    for (const auto& t : f.tensors) {
        const size_t dot = t.name.rfind('.');
        const std::string kind = dot == std::string::npos ? std::string() : t.name.substr(dot + 1);
        if (kind != "lora_a" && kind != "lora_b") { ad->error = "GGUF adapter tensor is neither .lora_a nor .lora_b: " + t.name; return false; }
        if ((t.type != 0 && t.type != 1) || !t.data || t.dims.size() != 2) { ad->error = "GGUF adapter tensors must be 2-d F32 or F16: " + t.name; return false; }
        Half h;
        h.data = t.data; h.f16 = t.type == 1; h.ne0 = (int64_t)t.dims[0]; h.ne1 = (int64_t)t.dims[1];
        (kind == "lora_a" ? as : bs)[t.name.substr(0, dot)] = h;
    }
    return pair_up(ad, as, bs, /*a_is_transposed=*/false); /* f's mapping is alive until here; pair_up copies */
}

bool TkLoraAdapter::load(const char* p) {
    tensors.clear();
    error.clear();
    path = p ? p : "";
    FILE* fp = p ? fopen(p, "rb") : nullptr;
    if (!fp) { error = std::string("cannot open LoRA adapter ") + path; return false; }
    uint8_t head[8] = {0};
    const size_t got = fread(head, 1, 8, fp);
    if (got < 8) { fclose(fp); error = "LoRA adapter file too small"; return false; }
    if (memcmp(head, "GGUF", 4) == 0) { fclose(fp); return load_gguf(this, p); }
    uint32_t magic, version;
    memcpy(&magic, head, 4);
    memcpy(&version, head + 4, 4);
    if (magic != 0x67676c61u) { fclose(fp); error = "not a LoRA adapter (neither ggla nor GGUF magic)"; return false; }
    if (version != 1) { fclose(fp); error = "unsupported ggla version"; return false; }
    fseek(fp, 0, SEEK_END);
    const long len = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (len < 16 || len > (1L << 32)) { fclose(fp); error = "ggla file of implausible size"; return false; }
    std::vector<uint8_t> buf((size_t)len);
    const bool ok = fread(buf.data(), 1, buf.size(), fp) == buf.size();
    fclose(fp);
    if (!ok) { error = "short read of the LoRA adapter"; return false; }
    return load_ggla(this, buf);
}
