/*
 * tk_tokenizer.h — prompt <-> token ids for the LLM runner.
 * The reference calls llama_tokenize(model, prompt, ..., add_bos=true, special=false)
 * (src/ai_models/tk_runner_streaming.c:24) and llama_token_to_piece (:84).  Two modes:
 *   - GGUF vocabulary present: llama-style SentencePiece BPE (score-ordered bigram merges,
 *     "▁" space marker with a leading space prefix, <0xXX> byte fallback) restated from the
 *     published algorithm;
 *   - synthetic checkpoints (no vocabulary): byte tokens, id = 3 + byte, BOS = 1, EOS = 2,
 *     which is exactly the byte-fallback range of the llama vocabulary.
 */
#ifndef TK_TOKENIZER_H
#define TK_TOKENIZER_H

#include <stdint.h>

#include <string>
#include <unordered_map>
#include <vector>

class TkTokenizer {
public:
    int bos = 1, eos = 2, unk = 0;
    int vocab = 0;
    void init_bytes(int vocab_size);
    void init_spm(const std::vector<std::string>& tokens, const std::vector<float>& scores, const std::vector<int32_t>& types, int bos_id,
                  int eos_id);
    std::vector<int32_t> encode(const std::string& text, bool add_bos) const;
    std::string piece(int32_t id) const;
    bool has_vocab() const { return !tokens_.empty(); }

private:
    std::vector<std::string> tokens_;
    std::vector<float> scores_;
    std::vector<int32_t> types_;
    std::unordered_map<std::string, int32_t> index_;
    int byte_token(uint8_t b) const;
};

#endif
