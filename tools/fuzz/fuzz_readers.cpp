// Mutation fuzzing of the host-side file readers under AddressSanitizer + UBSan (CPU build only; no HIP device involved):
//   LoRA adapters (ggla / GGUF), GGUF checkpoints, whisper.cpp ggml checkpoints, ONNX graphs and ONNX conv weights, GBNF grammars; the SentencePiece tokenizer on the vocabularies of accepted GGUF files.
// Every reader takes untrusted files from tk_*_create / tk_model_loader_load_model paths.  Build + run: tools/fuzz/run.sh
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "audio/tk_whisper_ggml.h"
#include "llm/tk_gguf.h"
#include "llm/tk_grammar.h"
#include "llm/tk_lora.h"
#include "llm/tk_tokenizer.h"
#include "nn/tk_onnx_graph.h"
#include "vision/tk_onnx_weights.h"

static std::vector<unsigned char> rd(const char* p) {
    std::vector<unsigned char> b;
    FILE* f = fopen(p, "rb");
    if (!f) return b;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    b.resize((size_t)n);
    if (fread(b.data(), 1, b.size(), f) != b.size()) b.clear();
    fclose(f);
    return b;
}
static void wr(const char* p, const std::vector<unsigned char>& b) {
    FILE* f = fopen(p, "wb");
    if (!b.empty()) fwrite(b.data(), 1, b.size(), f);
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: fuzz_readers SCRATCH_FILE ITERATIONS SAMPLE...\n"); return 2; }
    const char* scratch = argv[1];
    const int iters = atoi(argv[2]);
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    long ok = 0, bad = 0;
    for (int a = 3; a < argc; ++a) {
        const std::vector<unsigned char> base = rd(argv[a]);
        if (base.empty()) { fprintf(stderr, "cannot read %s\n", argv[a]); return 2; }
        for (int it = 0; it < iters; ++it) {
            std::vector<unsigned char> b = base;
            const int kind = (int)(rnd() % 4);
            if (kind == 0 && b.size() > 8) b.resize((size_t)(rnd() % b.size())); /* truncation */
            const int flips = 1 + (int)(rnd() % 8);
            for (int i = 0; i < flips && !b.empty(); ++i) { /* half of the damage in the header region, where the structure lives */
                const size_t pos = (rnd() % 2 == 0) ? rnd() % (b.size() < 2048 ? b.size() : 2048) : rnd() % b.size();
                if (kind == 2) b[pos] = 0xff; else if (kind == 3) b[pos] = 0; else b[pos] ^= (unsigned char)(1u << (rnd() % 8));
            }
            wr(scratch, b);
            { TkLoraAdapter r; (r.load(scratch) ? ok : bad)++; }
            {
                TkGgufFile r;
                if (r.open(scratch)) {
                    ++ok;
                    if (!r.tokens.empty()) { /* a vocabulary the file brought (damaged scores / types / ids included) must tokenise and render without faults */
                        TkTokenizer tok;
                        tok.init_spm(r.tokens, r.scores, r.token_type, (int)r.get("tokenizer.ggml.bos_token_id", 1), (int)r.get("tokenizer.ggml.eos_token_id", 2));
                        const std::vector<int32_t> ids = tok.encode(" hello world \xe2\x96\x81 \xff\x00x", true);
                        for (int32_t id : ids) (void)tok.piece(id);
                        (void)tok.piece(-1);
                        (void)tok.piece((int32_t)r.tokens.size());
                    }
                } else ++bad;
            }
            { TkWhisperGgml r; (r.open(scratch) ? ok : bad)++; }
            { TkOnnxGraph r; (r.load(scratch) ? ok : bad)++; }
            { TkOnnxWeights r; (r.load(scratch) ? ok : bad)++; }
            if (b.size() < 65536) { /* grammars are text: every small sample doubles as one */
                TkGrammar g;
                std::string err;
                const std::string text((const char*)b.data(), b.size());
                if (g.parse(text.c_str(), &err)) {
                    ++ok;
                    TkGrammarState st;
                    st.init(&g);
                    for (int k = 0; k < 64; ++k)
                        if (!st.accept((uint8_t)(rnd() & 0x7f))) break;
                } else ++bad;
            }
        }
    }
    printf("%ld inputs accepted, %ld rejected, no sanitizer report\n", ok, bad);
    return 0;
}
