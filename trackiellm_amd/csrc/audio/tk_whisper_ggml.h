/*
 * tk_whisper_ggml.h — reader of whisper.cpp's "ggml" checkpoint files (ggml-tiny.en.bin, …), the format the reference hands to
 * whisper_init_from_file_with_params (src/audio/tk_asr_whisper.c:238; model name in tests/tk_cortex_test.cpp:40-45).
 * whisper.cpp is not in the reference tree; the layout is the published one of its convert-pt-to-ggml.py:
 *
 *   u32 magic 0x67676d6c
 *   i32 n_vocab, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer, n_text_ctx, n_text_state, n_text_head, n_text_layer, n_mels, ftype
 *   i32 n_mel, i32 n_fft, f32 filters[n_mel][n_fft]
 *   i32 n_tokens, { u32 len, bytes[len] } x n_tokens                       (byte-level BPE pieces, already raw bytes)
 *   until EOF: i32 n_dims, i32 name_len, i32 type (0 f32, 1 f16), i32 ne[n_dims] (innermost first), name, data
 *
 * Tensor names are OpenAI's (encoder.blocks.0.attn.query.weight …) = the names of TkWhManifest.  Conv kernels are stored
 * [out][in][3] and are permuted to this path's [out][(tap, in)] rows; biases [n][1] flatten.  Quantised files (type >= 2) are
 * rejected.  Host-only code: parsing is tested without a GPU.
 */
#ifndef TK_WHISPER_GGML_H
#define TK_WHISPER_GGML_H

#include <stdint.h>

#include <string>
#include <vector>

#include "../common/tk_whisper_graph.h"

struct TkWhisperGgmlTensor {
    std::string name;
    int n_dims = 0;
    int64_t ne[4] = {1, 1, 1, 1}; /* innermost first, as stored */
    int type = 0;                 /* 0 f32, 1 f16 */
    int64_t offset = 0;           /* of the data in the file */
    int64_t count = 0;            /* elements */
};

class TkWhisperGgml {
public:
    TkWhisperHP hp{};
    int ftype = 0;
    std::vector<float> mel_filters; /* [n_mels][201] */
    std::vector<std::string> vocab; /* token id -> bytes */
    std::vector<TkWhisperGgmlTensor> tensors;
    std::string error;

    static bool is_ggml(const char* path); /* magic check only */
    bool open(const char* path);           /* header, filters, vocabulary, tensor directory (data is not read) */
    bool open_checked(const char* path);   /* open() without the exception barrier */
    /* tensor `idx` of `man` as fp32 rows x cols in this path's layout; frontend tables (hann / DFT) are not in the file -> false with found = false */
    bool read(const TkWhManifest& man, int idx, std::vector<float>* out, bool* found);

private:
    std::string path_;
};

#endif
