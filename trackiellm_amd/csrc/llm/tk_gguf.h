/*
 * tk_gguf.h — minimal GGUF v2/v3 reader (metadata + tensor directory, data is mmap'd).
 * The reference never parses the file itself: it hands the path to
 * llama_load_model_from_file (src/ai_models/tk_model_loader.c:245-251) and reads a few
 * metadata keys back (:780-806).  Format restated from the public GGUF specification
 * (ggml-org/ggml docs/gguf.md), third-party and un-vendored.
 */
#ifndef TK_GGUF_H
#define TK_GGUF_H

#include <stdint.h>

#include <map>
#include <string>
#include <vector>

struct TkGgufTensor {
    std::string name;
    std::vector<uint64_t> dims; /* ne[0] is the contiguous (K) dimension */
    uint32_t type = 0;
    uint64_t offset = 0; /* relative to data section */
    const uint8_t* data = nullptr;
    size_t nbytes = 0;
};

struct TkGgufFile {
    std::string error;
    uint32_t version = 0;
    std::map<std::string, double> num;          /* every scalar numeric / bool KV */
    std::map<std::string, std::string> str;     /* every string KV */
    std::vector<std::string> tokens;            /* tokenizer.ggml.tokens */
    std::vector<float> scores;                  /* tokenizer.ggml.scores */
    std::vector<int32_t> token_type;            /* tokenizer.ggml.token_type */
    std::vector<TkGgufTensor> tensors;

    ~TkGgufFile();
    bool open(const char* path);
    const TkGgufTensor* find(const std::string& name) const;
    double get(const std::string& key, double dflt) const;

private:
    bool open_checked(const char* path);
    void* map_ = nullptr;
    size_t map_len_ = 0;
};

#endif
