/* tk_onnx_graph.cpp — see tk_onnx_graph.h */
#include "tk_onnx_graph.h"

#include <stdio.h>
#include <string.h>

#include <stdexcept>

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

std::string str(Span s) { return std::string((const char*)s.p, (size_t)(s.e - s.p)); }

bool parse_tensor(Span s, std::string* name, TkOnnxTensor* t) {
    Span raw{nullptr, nullptr};
    uint32_t num, wt;
    Span sub;
    uint64_t v;
    std::vector<float> fdata;
    std::vector<int64_t> idata;
    while (s.p < s.e) {
        if (!field(s, &num, &wt, &sub, &v)) return false;
        if (num == 1) {
            if (wt == 0) t->dims.push_back((int64_t)v);
            else if (wt == 2) { uint64_t d; while (sub.p < sub.e) { if (!varint(sub, &d)) return false; t->dims.push_back((int64_t)d); } }
        } else if (num == 2 && wt == 0) t->dtype = (int)v;
        else if (num == 4) {
            if (wt == 5) { float f; uint32_t u = (uint32_t)v; memcpy(&f, &u, 4); fdata.push_back(f); }
            else if (wt == 2) { const size_t n = (size_t)(sub.e - sub.p) / 4, o = fdata.size(); fdata.resize(o + n); if (n) memcpy(fdata.data() + o, sub.p, n * 4); }
        } else if (num == 5 || num == 7) { /* int32_data / int64_data: varints, packed or not */
            if (wt == 0) idata.push_back((int64_t)v);
            else if (wt == 2) { uint64_t d; while (sub.p < sub.e) { if (!varint(sub, &d)) return false; idata.push_back((int64_t)d); } }
        } else if (num == 8 && wt == 2) *name = str(sub);
        else if (num == 9 && wt == 2) raw = sub;
    }
    int64_t count = 1;
    for (int64_t d : t->dims) { if (d < 0 || d > (1 << 26)) return false; count *= d; if (count > (1 << 28)) return false; }
    const int64_t rawn = raw.p ? (int64_t)(raw.e - raw.p) : -1;
    if (t->dtype == 1) {
        if (rawn == count * 4) { t->f.resize((size_t)count); if (count) memcpy(t->f.data(), raw.p, (size_t)count * 4); } /* an empty tensor has no bytes to copy (and no buffer to copy into) */
        else if ((int64_t)fdata.size() == count) t->f.swap(fdata);
        else return count == 0;
    } else if (t->dtype == 10) {
        if (rawn != count * 2) return false;
        t->f.resize((size_t)count);
        for (int64_t i = 0; i < count; ++i) { uint16_t h; memcpy(&h, raw.p + 2 * i, 2); t->f[(size_t)i] = tk_f16_to_f32(h); }
    } else if (t->dtype == 7) {
        if (rawn == count * 8) { t->i.resize((size_t)count); if (count) memcpy(t->i.data(), raw.p, (size_t)count * 8); }
        else if ((int64_t)idata.size() == count) t->i.swap(idata);
        else return count == 0;
    } else if (t->dtype == 6) {
        if (rawn == count * 4) { t->i.resize((size_t)count); for (int64_t i = 0; i < count; ++i) { int32_t x; memcpy(&x, raw.p + 4 * i, 4); t->i[(size_t)i] = x; } }
        else if ((int64_t)idata.size() == count) { t->i.resize((size_t)count); for (int64_t i = 0; i < count; ++i) t->i[(size_t)i] = (int32_t)idata[(size_t)i]; }
        else return count == 0;
    } else if (t->dtype == 9) { /* bool (attention masks of Where nodes): one byte per element in raw_data, or int32_data */
        if (rawn == count) { t->i.resize((size_t)count); for (int64_t i = 0; i < count; ++i) t->i[(size_t)i] = raw.p[i] ? 1 : 0; }
        else if ((int64_t)idata.size() == count) { t->i.resize((size_t)count); for (int64_t i = 0; i < count; ++i) t->i[(size_t)i] = idata[(size_t)i] ? 1 : 0; }
        else return count == 0;
    } /* other types (double, ...) are kept as shape-only entries: a node that needs them fails by name */
    return true;
}

bool parse_graph(Span g, TkOnnxGraph* out, int depth, std::string* err);

bool parse_attr(Span s, std::string* name, TkOnnxAttr* a, int depth) {
    uint32_t num, wt;
    Span sub;
    uint64_t v;
    while (s.p < s.e) {
        if (!field(s, &num, &wt, &sub, &v)) return false;
        if (num == 1 && wt == 2) *name = str(sub);
        else if (num == 2 && wt == 5) { uint32_t u = (uint32_t)v; memcpy(&a->f, &u, 4); }
        else if (num == 3 && wt == 0) a->i = (int64_t)v;
        else if (num == 4 && wt == 2) a->s = str(sub);
        else if (num == 5 && wt == 2) { std::string tn; if (!parse_tensor(sub, &tn, &a->t)) return false; a->has_t = true; }
        else if (num == 6 && wt == 2) {
            if (depth >= 4) return false; /* If inside If inside ...: four levels are more than any model of these classes nests */
            a->g = std::make_shared<TkOnnxGraph>();
            std::string err;
            if (!parse_graph(sub, a->g.get(), depth + 1, &err)) return false;
        } else if (num == 7) {
            if (wt == 5) { float f; uint32_t u = (uint32_t)v; memcpy(&f, &u, 4); a->floats.push_back(f); }
            else if (wt == 2) { const size_t n = (size_t)(sub.e - sub.p) / 4, o = a->floats.size(); a->floats.resize(o + n); memcpy(a->floats.data() + o, sub.p, n * 4); }
        } else if (num == 8) {
            if (wt == 0) a->ints.push_back((int64_t)v);
            else if (wt == 2) { uint64_t d; while (sub.p < sub.e) { if (!varint(sub, &d)) return false; a->ints.push_back((int64_t)d); } }
        }
    }
    return true;
}

bool parse_value_info(Span s, TkOnnxValueInfo* vi) {
    uint32_t num, wt;
    Span sub, type{nullptr, nullptr};
    uint64_t v;
    while (s.p < s.e) {
        if (!field(s, &num, &wt, &sub, &v)) return false;
        if (num == 1 && wt == 2) vi->name = str(sub);
        else if (num == 2 && wt == 2) type = sub;
    }
    if (!type.p) return true;
    Span tt{nullptr, nullptr};
    while (type.p < type.e) {
        if (!field(type, &num, &wt, &sub, &v)) return false;
        if (num == 1 && wt == 2) tt = sub;
    }
    if (!tt.p) return true;
    Span shape{nullptr, nullptr};
    while (tt.p < tt.e) {
        if (!field(tt, &num, &wt, &sub, &v)) return false;
        if (num == 1 && wt == 0) vi->elem_type = (int)v;
        else if (num == 2 && wt == 2) shape = sub;
    }
    while (shape.p && shape.p < shape.e) {
        if (!field(shape, &num, &wt, &sub, &v)) return false;
        if (num == 1 && wt == 2) {
            int64_t dv = -1;
            Span d = sub, s2;
            uint32_t n2, w2;
            uint64_t v2;
            while (d.p < d.e) {
                if (!field(d, &n2, &w2, &s2, &v2)) return false;
                if (n2 == 1 && w2 == 0) dv = (int64_t)v2;
            }
            vi->dims.push_back(dv);
        }
    }
    return true;
}

bool parse_graph(Span g, TkOnnxGraph* out, int depth, std::string* err) {
    uint32_t num, wt;
    Span sub;
    uint64_t v;
    std::vector<TkOnnxValueInfo> all_in;
    while (g.p < g.e) {
        if (!field(g, &num, &wt, &sub, &v)) { *err = "corrupt GraphProto"; return false; }
        if (num == 1 && wt == 2) {
            TkOnnxNode nd;
            Span ns = sub, f2;
            uint32_t fn, fw;
            uint64_t fv;
            while (ns.p < ns.e) {
                if (!field(ns, &fn, &fw, &f2, &fv)) { *err = "corrupt NodeProto"; return false; }
                if (fn == 1 && fw == 2) nd.in.push_back(str(f2));
                else if (fn == 2 && fw == 2) nd.out.push_back(str(f2));
                else if (fn == 3 && fw == 2) nd.name = str(f2);
                else if (fn == 4 && fw == 2) nd.op = str(f2);
                else if (fn == 5 && fw == 2) {
                    std::string an;
                    TkOnnxAttr a;
                    if (!parse_attr(f2, &an, &a, depth)) { *err = "corrupt AttributeProto in node " + nd.name; return false; }
                    nd.attr[an] = std::move(a);
                }
            }
            out->nodes.push_back(std::move(nd));
        } else if (num == 5 && wt == 2) {
            std::string name;
            TkOnnxTensor t;
            if (!parse_tensor(sub, &name, &t)) { *err = "corrupt or oversized initialiser " + name; return false; }
            out->init[name] = std::move(t);
        } else if ((num == 11 || num == 12) && wt == 2) {
            TkOnnxValueInfo vi;
            if (!parse_value_info(sub, &vi)) { *err = "corrupt ValueInfoProto"; return false; }
            (num == 11 ? all_in : out->outputs).push_back(std::move(vi));
        }
    }
    for (auto& vi : all_in)
        if (!out->init.count(vi.name)) out->inputs.push_back(vi); /* old exporters list initialisers among the inputs */
    return true;
}
}  // namespace

bool TkOnnxGraph::load(const char* path) {
    try {
        nodes.clear(); init.clear(); inputs.clear(); outputs.clear();
        FILE* f = fopen(path, "rb");
        if (!f) { error = std::string("cannot open ") + path; return false; }
        fseek(f, 0, SEEK_END);
        const long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (n <= 0 || n > (1L << 30)) { fclose(f); error = "unreasonable file size"; return false; }
        std::vector<uint8_t> buf((size_t)n);
        const bool rd = fread(buf.data(), 1, (size_t)n, f) == (size_t)n;
        fclose(f);
        if (!rd) { error = "read error"; return false; }
        Span m{buf.data(), buf.data() + buf.size()}, graph{nullptr, nullptr};
        uint32_t num, wt;
        Span sub;
        uint64_t v;
        while (m.p < m.e) {
            if (!field(m, &num, &wt, &sub, &v)) { error = "corrupt ModelProto"; return false; }
            if (num == 7 && wt == 2) graph = sub;
        }
        if (!graph.p) { error = "no graph in the ONNX file"; return false; }
        if (!parse_graph(graph, this, 0, &error)) return false;
        if (nodes.empty()) { error = "the ONNX graph has no nodes"; return false; }
        return true;
    } catch (const std::exception& e) {
        error = std::string("corrupt ONNX file (") + e.what() + ")";
        return false;
    }
}

void TkOnnxGraph::all_nodes(std::vector<const TkOnnxNode*>* out) const {
    for (const auto& nd : nodes) {
        out->push_back(&nd);
        for (const auto& kv : nd.attr)
            if (kv.second.g) kv.second.g->all_nodes(out);
    }
}
