#include "tk_gguf.h"

#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <stdexcept>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../common/tk_ggml_blocks.h"

namespace {
struct Cursor {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    template <typename T> T rd() {
        T v{};
        if (p + sizeof(T) > end) { ok = false; return v; }
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    std::string rds() {
        uint64_t n = rd<uint64_t>();
        if (!ok || n > (uint64_t)(end - p)) { ok = false; return std::string(); }
        std::string s((const char*)p, (size_t)n);
        p += n;
        return s;
    }
};

size_t scalar_size(uint32_t t) {
    switch (t) {
        case 0: case 1: case 7: return 1;
        case 2: case 3: return 2;
        case 4: case 5: case 6: return 4;
        case 10: case 11: case 12: return 8;
        default: return 0;
    }
}

double rd_scalar(Cursor& c, uint32_t t) {
    switch (t) {
        case 0: return c.rd<uint8_t>();
        case 1: return c.rd<int8_t>();
        case 2: return c.rd<uint16_t>();
        case 3: return c.rd<int16_t>();
        case 4: return c.rd<uint32_t>();
        case 5: return c.rd<int32_t>();
        case 6: return c.rd<float>();
        case 7: return c.rd<uint8_t>() ? 1.0 : 0.0;
        case 10: return (double)c.rd<uint64_t>();
        case 11: return (double)c.rd<int64_t>();
        case 12: return c.rd<double>();
        default: c.ok = false; return 0.0;
    }
}
}  // namespace

TkGgufFile::~TkGgufFile() {
    if (map_) munmap(map_, map_len_);
}

double TkGgufFile::get(const std::string& key, double dflt) const {
    auto it = num.find(key);
    return it == num.end() ? dflt : it->second;
}

const TkGgufTensor* TkGgufFile::find(const std::string& name) const {
    for (const auto& t : tensors)
        if (t.name == name) return &t;
    return nullptr;
}

bool TkGgufFile::open(const char* path) {
    /* file contents are untrusted: nothing below may throw through the extern "C" callers */
    try {
        return open_checked(path);
    } catch (const std::exception& e) {
        error = std::string("corrupt GGUF file (") + e.what() + ")";
        return false;
    }
}

bool TkGgufFile::open_checked(const char* path) {
    int fd = ::open(path, O_RDONLY);
    if (fd < 0) { error = std::string("cannot open ") + path; return false; }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 24) { ::close(fd); error = "file too small to be GGUF"; return false; }
    map_len_ = (size_t)st.st_size;
    map_ = mmap(nullptr, map_len_, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (map_ == MAP_FAILED) { map_ = nullptr; error = "mmap failed"; return false; }
    Cursor c{(const uint8_t*)map_, (const uint8_t*)map_ + map_len_};
    if (memcmp(c.p, "GGUF", 4) != 0) { error = "bad magic (not a GGUF file)"; return false; }
    c.p += 4;
    version = c.rd<uint32_t>();
    if (version < 2 || version > 3) { error = "unsupported GGUF version"; return false; }
    uint64_t n_tensors = c.rd<uint64_t>();
    uint64_t n_kv = c.rd<uint64_t>();
    if (!c.ok || n_tensors > (1u << 20) || n_kv > (1u << 20)) { error = "corrupt GGUF header"; return false; }
    for (uint64_t i = 0; i < n_kv && c.ok; ++i) {
        std::string key = c.rds();
        uint32_t t = c.rd<uint32_t>();
        if (t == 8) {
            str[key] = c.rds();
        } else if (t == 9) {
            uint32_t et = c.rd<uint32_t>();
            uint64_t n = c.rd<uint64_t>();
            if (!c.ok) break;
            if (et == 8) {
                /* every string costs at least its 8-byte length field: a count the remaining bytes cannot hold is corruption, not a
                 * reason to reserve() whatever a crafted file asks for */
                if (n > (uint64_t)(c.end - c.p) / 8) { c.ok = false; break; }
                std::vector<std::string> v;
                v.reserve((size_t)n);
                for (uint64_t k = 0; k < n && c.ok; ++k) v.push_back(c.rds());
                if (key == "tokenizer.ggml.tokens") tokens.swap(v);
            } else {
                size_t es = scalar_size(et);
                if (es == 0 || n > (uint64_t)(c.end - c.p) / es) { c.ok = false; break; }
                if (key == "tokenizer.ggml.scores" && et == 6) {
                    scores.resize((size_t)n);
                    memcpy(scores.data(), c.p, (size_t)n * 4);
                } else if (key == "tokenizer.ggml.token_type" && et == 5) {
                    token_type.resize((size_t)n);
                    memcpy(token_type.data(), c.p, (size_t)n * 4);
                }
                c.p += n * es;
            }
        } else {
            num[key] = rd_scalar(c, t);
        }
    }
    if (!c.ok) { error = "corrupt GGUF metadata"; return false; }
    tensors.resize((size_t)n_tensors);
    for (auto& t : tensors) {
        t.name = c.rds();
        uint32_t nd = c.rd<uint32_t>();
        if (!c.ok || nd > 4) { error = "corrupt GGUF tensor directory"; return false; }
        t.dims.resize(nd);
        for (auto& d : t.dims) d = c.rd<uint64_t>();
        t.type = c.rd<uint32_t>();
        t.offset = c.rd<uint64_t>();
    }
    if (!c.ok) { error = "corrupt GGUF tensor directory"; return false; }
    uint64_t align = (uint64_t)get("general.alignment", 32);
    if (align == 0) align = 32;
    uint64_t pos = (uint64_t)(c.p - (const uint8_t*)map_);
    if (align > (1u << 20) || pos > map_len_) { error = "corrupt GGUF alignment"; return false; }
    uint64_t data0 = (pos + align - 1) / align * align; /* pos <= file size and align <= 2^20: no wrap */
    if (data0 > map_len_) {
        /* a vocabulary-only file (llama.cpp ships such ggml-vocab-*.gguf) may end before the padding of an empty data section */
        if (!tensors.empty()) { error = "GGUF tensor data section is missing"; return false; }
        data0 = map_len_;
    }
    const uint64_t data_len = map_len_ - data0;
    for (auto& t : tensors) {
        uint64_t n = 1;
        bool overflow = false;
        for (auto d : t.dims) {
            if (d != 0 && n > UINT64_MAX / d) { overflow = true; break; }
            n *= d;
        }
        if (overflow) { error = "tensor element count overflows: " + t.name; return false; }
        size_t be = tk_type_block_elems((int)t.type), bb = tk_type_block_bytes((int)t.type);
        if (t.type != TK_TYPE_F32 && t.type != TK_TYPE_F16 && t.type != TK_TYPE_Q4_K && t.type != TK_TYPE_Q6_K) {
            t.nbytes = 0; /* unsupported type: reported when a consumer asks for this tensor */
            t.data = nullptr;
            continue;
        }
        if (n / be > data_len / bb) { error = "tensor data runs past end of file: " + t.name; return false; } /* also rules out n / be * bb wrapping */
        t.nbytes = (size_t)(n / be * bb);
        if (t.offset > data_len || t.nbytes > data_len - t.offset) { error = "tensor data runs past end of file: " + t.name; return false; }
        t.data = (const uint8_t*)map_ + data0 + t.offset;
    }
    return true;
}
