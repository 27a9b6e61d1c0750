/* tk_onnx_weights.cpp — see tk_onnx_weights.h */
#include "tk_onnx_weights.h"

#include <stdio.h>
#include <string.h>

#include <map>

#include "../common/tk_exact_math.h"

namespace {

struct Span { const uint8_t* p; const uint8_t* e; };

bool varint(Span& s, uint64_t* v) {
    uint64_t r = 0;
    for (int sh = 0; sh < 64 && s.p < s.e; sh += 7) {
        const uint8_t b = *s.p++;
        r |= (uint64_t)(b & 0x7f) << sh;
        if (!(b & 0x80)) { *v = r; return true; }
    }
    return false;
}

/* next field of a message: number, wire type, and for length-delimited fields its bytes; scalar value in *val */
bool field(Span& s, uint32_t* num, uint32_t* wt, Span* sub, uint64_t* val) {
    uint64_t key;
    if (!varint(s, &key)) return false;
    *num = (uint32_t)(key >> 3);
    *wt = (uint32_t)(key & 7);
    switch (*wt) {
        case 0: return varint(s, val);
        case 1: if (s.e - s.p < 8) return false; memcpy(val, s.p, 8); s.p += 8; return true;
        case 5: if (s.e - s.p < 4) return false; *val = 0; memcpy(val, s.p, 4); s.p += 4; return true;
        case 2: {
            uint64_t n;
            if (!varint(s, &n) || n > (uint64_t)(s.e - s.p)) return false;
            sub->p = s.p; sub->e = s.p + n; s.p += n;
            return true;
        }
        default: return false;
    }
}

struct Tensor { std::vector<int64_t> dims; int dtype = 0; std::vector<float> data; };

bool parse_tensor(Span s, std::string* name, Tensor* t, std::string* err) {
    Span raw{nullptr, nullptr};
    std::vector<float> fdata;
    uint32_t num, wt;
    Span sub;
    uint64_t v;
    while (s.p < s.e) {
        if (!field(s, &num, &wt, &sub, &v)) { *err = "corrupt TensorProto"; return false; }
        if (num == 1) { /* dims: packed or repeated varints */
            if (wt == 0) t->dims.push_back((int64_t)v);
            else if (wt == 2) { uint64_t d; while (sub.p < sub.e) { if (!varint(sub, &d)) return false; t->dims.push_back((int64_t)d); } }
        } else if (num == 2 && wt == 0) t->dtype = (int)v;
        else if (num == 4) { /* float_data */
            if (wt == 5) { float f; uint32_t u = (uint32_t)v; memcpy(&f, &u, 4); fdata.push_back(f); }
            else if (wt == 2) { const size_t n = (size_t)(sub.e - sub.p) / 4; const size_t o = fdata.size(); fdata.resize(o + n); if (n) memcpy(fdata.data() + o, sub.p, n * 4); }
        } else if (num == 8 && wt == 2) name->assign((const char*)sub.p, (size_t)(sub.e - sub.p));
        else if (num == 9 && wt == 2) raw = sub;
    }
    int64_t count = 1;
    for (int64_t d : t->dims) { if (d <= 0 || d > (1 << 24)) { *err = "bad tensor dims"; return false; } count *= d; }
    if (t->dtype == 1) {
        if (raw.p) { if ((int64_t)(raw.e - raw.p) != count * 4) return true; /* not a weight we can use */ t->data.resize((size_t)count); memcpy(t->data.data(), raw.p, (size_t)count * 4); }
        else if ((int64_t)fdata.size() == count) t->data.swap(fdata);
    } else if (t->dtype == 10 && raw.p && (int64_t)(raw.e - raw.p) == count * 2) {
        t->data.resize((size_t)count);
        for (int64_t i = 0; i < count; ++i) { uint16_t h; memcpy(&h, raw.p + 2 * i, 2); t->data[(size_t)i] = tk_f16_to_f32(h); }
    }
    return true;
}

bool read_all(const char* path, std::vector<uint8_t>* buf, std::string* err) {
    FILE* f = fopen(path, "rb");
    if (!f) { *err = std::string("cannot open ") + path; return false; }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n <= 0 || n > (1L << 31)) { fclose(f); *err = "unreasonable file size"; return false; }
    buf->resize((size_t)n);
    const bool ok = fread(buf->data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    if (!ok) *err = "read error";
    return ok;
}

}  // namespace

bool TkOnnxWeights::looks_like_onnx(const char* path) {
    std::vector<uint8_t> buf;
    std::string e;
    if (!read_all(path, &buf, &e)) return false;
    Span s{buf.data(), buf.data() + buf.size()};
    uint32_t num, wt;
    Span sub;
    uint64_t v;
    while (s.p < s.e) {
        if (!field(s, &num, &wt, &sub, &v)) return false;
        if (num == 7 && wt == 2) return true;
        if (num > 30) return false;
    }
    return false;
}

bool TkOnnxWeights::load(const char* path) {
    convs.clear();
    std::vector<uint8_t> buf;
    if (!read_all(path, &buf, &error)) return false;
    Span m{buf.data(), buf.data() + buf.size()}, graph{nullptr, nullptr};
    uint32_t num, wt;
    Span sub;
    uint64_t v;
    while (m.p < m.e) {
        if (!field(m, &num, &wt, &sub, &v)) { error = "corrupt ModelProto"; return false; }
        if (num == 7 && wt == 2) graph = sub;
    }
    if (!graph.p) { error = "no graph in the ONNX file"; return false; }
    std::map<std::string, Tensor> init;
    std::vector<std::vector<std::string>> conv_inputs;
    Span g = graph;
    while (g.p < g.e) {
        if (!field(g, &num, &wt, &sub, &v)) { error = "corrupt GraphProto"; return false; }
        if (num == 5 && wt == 2) {
            std::string name;
            Tensor t;
            if (!parse_tensor(sub, &name, &t, &error)) return false;
            if (!t.data.empty()) init[name] = std::move(t);
        } else if (num == 1 && wt == 2) {
            std::vector<std::string> ins;
            std::string op;
            Span n = sub, f2;
            uint32_t fn, fw;
            uint64_t fv;
            while (n.p < n.e) {
                if (!field(n, &fn, &fw, &f2, &fv)) { error = "corrupt NodeProto"; return false; }
                if (fn == 1 && fw == 2) ins.emplace_back((const char*)f2.p, (size_t)(f2.e - f2.p));
                else if (fn == 4 && fw == 2) op.assign((const char*)f2.p, (size_t)(f2.e - f2.p));
            }
            if (op == "Conv") conv_inputs.push_back(ins);
        }
    }
    for (const auto& ins : conv_inputs) {
        if (ins.size() < 2) { error = "Conv node without a weight input"; return false; }
        auto wi = init.find(ins[1]);
        if (wi == init.end() || wi->second.dims.size() != 4) { error = "Conv weight '" + ins[1] + "' is not a 4-D float initialiser"; return false; }
        TkOnnxConv c;
        c.cout = (int)wi->second.dims[0]; c.cin = (int)wi->second.dims[1]; c.kh = (int)wi->second.dims[2]; c.kw = (int)wi->second.dims[3];
        c.w = wi->second.data;
        c.b.assign((size_t)c.cout, 0.0f);
        if (ins.size() >= 3 && !ins[2].empty()) {
            auto bi = init.find(ins[2]);
            if (bi == init.end() || (int64_t)bi->second.data.size() != c.cout) { error = "Conv bias '" + ins[2] + "' is missing or has the wrong size"; return false; }
            c.b = bi->second.data;
        }
        convs.push_back(std::move(c));
    }
    if (convs.empty()) { error = "the ONNX graph has no Conv nodes"; return false; }
    return true;
}
