/*
 * tk_sample.h — the sampling state shared by the LLM stream's k_argmax and the ASR decoder's token pick (host + device).
 */
#ifndef TK_SAMPLE_H
#define TK_SAMPLE_H

#include <stdint.h>

/* per-row sampling state: temp <= 0 = greedy (arg max); otherwise the reference's default chain (top-k, top-p, min-p, temperature, one draw
 * from a counter-based generator); the kernel adds one to `counter` per sampled token.  Vocabularies up to 65536 tokens. */
#define TK_SAMPLE_MAX_K 64
struct TkSampleRow {
    float temp, top_p, min_p;
    int32_t top_k;
    uint64_t seed;
    uint32_t counter, pad;
};

#endif
