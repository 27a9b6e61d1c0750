/*
 * tk_audio_engine.h — ASR + VAD streams on the GPU.
 *   TkWhisperModel / TkAsr  <->  whisper_init_from_file_with_params + whisper_full  (src/audio/tk_asr_whisper.c:238,142-147)
 *   TkVadModel              <->  the Silero ONNX session run per 30 ms window       (src/sensors/tk_vad_silero.c:193-280)
 */
#ifndef TK_AUDIO_ENGINE_H
#define TK_AUDIO_ENGINE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../common/tk_whisper_graph.h"

class TkWhisperModel {
public:
    TkWhisperHP hp{};
    TkWhManifest man;
    int device = 0;
    std::vector<float*> w; /* device tensors in manifest order */
    /* the linear layers' weights once more as the tiled GEMM's f32 weight tiles (csrc/nn/tk_gemm_tiled.h), null where a tensor is not a
     * linear weight or its K is not a multiple of 128; built by prepare_tiles(), refreshed by set_tensor() */
    std::vector<uint8_t*> wt;
    bool prepare_tiles();
    int tensor_of(const float* dev_ptr) const; /* manifest index of a device tensor, -1 when it is none */
    std::string error;
    ~TkWhisperModel();
    bool init(const TkWhisperHP& hp, int device);
    bool fill_synthetic(uint64_t seed);
    bool set_tensor(int idx, const float* host, size_t n);
    bool load_file(const char* path); /* "TKWHSP1\0" container */
    /* whisper.cpp ggml checkpoint (f16 / f32), already opened: every manifest tensor must be present except the hann / DFT tables */
    bool load_ggml(class TkWhisperGgml& g);
};

class TkAsr {
public:
    TkWhisperModel* model = nullptr;
    int max_batch = 1;
    /* opt-in: the contractions of long passes (log-mel; encoder: convolutions, linear layers, attention as ONE kernel: k_attention_h3) on the f16
     * matrix pipe with split operands (TkGemm::fast) — ~1e-6 of scale off the exact chains; decoder steps (a few rows) keep the exact tiled path */
    bool fast = false;
    std::string error;
    hipStream_t stream = nullptr;
    ~TkAsr();
    bool init(TkWhisperModel* m, int max_batch);
    /* pcm[B][n_samples] (host), greedy decode of exactly n_steps tokens after the start-of-transcript prompt;
     * tokens_out[B][n_steps]; optional copies of the mel / encoder output for parity tests */
    bool transcribe(int B, const int16_t* pcm, int n_samples, const int32_t* prompt, int n_prompt, int n_steps, int32_t* tokens_out,
                    std::vector<float>* mel_out, std::vector<float>* enc_out, std::vector<float>* first_logits);
    /* the same with whisper.cpp's per-step bookkeeping: temperature 0 = arg max, > 0 = one draw per step from softmax(l / temperature) by the
     * canonical sampler keyed by (seed, step x B + b); logprobs_out [B][n_steps] = log-probability of every produced token under that
     * distribution (what the policy's logprob / entropy thresholds are taken on) */
    bool transcribe_policy(int B, const int16_t* pcm, int n_samples, const int32_t* prompt, int n_prompt, int n_steps, float temperature, uint64_t seed,
                           int32_t* tokens_out, float* logprobs_out);
    /* the decode as whisper.cpp runs it under the reference's parameters (src/audio/tk_asr_whisper.c:89-110; csrc/nn/tk_nn_kernels.h: TkWhFilter):
     * logit filters, timestamp rules and the per-token bookkeeping on the device, at most n_steps tokens per utterance.  n_samples_row [B] = each
     * utterance's own length (rows are padded to `n_samples` with zeros): it sets the row's seek_end.  suppress [n_vocab] = 1 where a token is
     * never sampled.  Out: tokens [B][n_steps] (eot behind a row's end), logprobs [B][n_steps] (may be null), result_len [B] = tokens that make
     * up the row's text, status [B] = 0 (n_steps reached), 1 (completed), 2 (failed: whisper.cpp would fall back to the next temperature) */
    bool transcribe_ref(int B, const int16_t* pcm, int n_samples, const int32_t* n_samples_row, const int32_t* prompt, int n_prompt, int n_steps, float temperature,
                        uint64_t seed, const uint8_t* suppress, int32_t token_beg, int32_t token_eot, int32_t* tokens_out, float* logprobs_out, int32_t* result_len,
                        int32_t* status);

private:
    friend struct TkAudioGpuOps;
    float* arena = nullptr;
    size_t arena_floats = 0, arena_used = 0;
    std::string launch_error; /* a launcher refused its arguments while the graph was being enqueued (no HIP error is raised for that) */
    int16_t* pcm_dev = nullptr;
    size_t pcm_cap = 0;
    struct { bool on = false; float temp = 0.0f; uint64_t seed = 0; float* logprob = nullptr; int step = 0;
             bool filtered = false; int first_step = 0; const uint8_t* suppress = nullptr; int32_t* state = nullptr; int32_t beg = 0, eot = 0, tid0 = 0; } pick; /* token pick of the current transcribe */
    uint8_t* suppress_dev = nullptr; /* [n_vocab] of the last transcribe_ref (uploaded again only when the table changes) */
    std::vector<uint8_t> suppress_host;
    int32_t* wh_state = nullptr;     /* [max_batch][TK_WH_STATE_INTS] */
    float* lp_dev = nullptr;         /* [n_text_ctx][max_batch] log-probabilities of a policy / reference-parameter decode */
    bool ensure_decode_buffers();
    std::vector<float> pick_logprobs; /* [total steps][B], read back by transcribe_policy */
};

/* tiny MLP speech-probability model over one 30 ms window: sigmoid(w2 . relu(W1 x + b1) + b2) */
class TkVadModel {
public:
    int device = 0, window = 480, hidden = 64;
    float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr;
    float *x = nullptr, *hbuf = nullptr, *p = nullptr; /* grow-only scratch */
    int cap = 0;
    hipStream_t stream = nullptr;
    std::string error;
    ~TkVadModel();
    bool init(int device, int window, int hidden);
    bool fill_synthetic(uint64_t seed);
    /* windows: [n][window] float on the host -> probabilities[n] */
    bool infer(const float* windows_host, int n, float* prob_host);
};

#endif
