/* C host for bench.py's reference_abi_batched leg: K tk_llm_runner_t handles on ONE model handle, each driven by its own pthread
 * through tk_model_loader_load_model / tk_llm_runner_prepare_generation / tk_llm_runner_generate_next_token only — the way a C host of
 * the reference would (src/cortex/tk_cortex_main.c:1323-1379), with no interpreter in the loop.  Same workload as the Python driver in
 * bench.py (64-token prompts, N tokens each, two rounds, the second one timed), so the difference between the two is the driver's cost.
 *   cc -O2 -std=c11 -Iinclude tools/abi_runners_host.c -Ltrackiellm_amd -ltrackie_mi355x -lpthread -o build/abi_runners_host
 *   build/abi_runners_host K N   ->  one JSON line */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "tk/tk_mi355x_ext.h"
#include "tk/tk_model_runner.h"
#include "tk/tk_types.h"

typedef struct { tk_llm_runner_t* runner; char prompt[64]; int n_tokens, count, rc; } job_t;

static void* drive(void* arg) {
    job_t* j = (job_t*)arg;
    j->count = 0;
    j->rc = (int)tk_llm_runner_prepare_generation(j->runner, j->prompt, false);
    if (j->rc != 0) return NULL;
    for (int i = 0; i < j->n_tokens; ++i) {
        if (!tk_llm_runner_generate_next_token(j->runner)) break;
        j->count++;
    }
    return NULL;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 16, N = argc > 2 ? atoi(argv[2]) : 128;
    if (K < 1 || K > 256 || N < 1) return 2;
    tk_model_loader_t* loader = NULL;
    tk_model_loader_config_t lc = {2, 1};
    if (tk_model_loader_create(&loader, &lc) != TK_SUCCESS) return 3;
    tk_path_t* path = NULL;
    if (tk_path_create_from_string(&path, "synthetic://mistral-7b?seed=4") != TK_SUCCESS) return 4;
    tk_model_load_params_t lp;
    memset(&lp, 0, sizeof lp);
    lp.model_path = path; lp.model_type = TK_MODEL_FORMAT_GGUF; lp.gpu_layers = 99;
    void* model = NULL;
    if (tk_model_loader_load_model(loader, &lp, &model) != TK_SUCCESS) { fprintf(stderr, "load: %s\n", tk_error_get_detail()); return 5; }
    tk_path_destroy(&path);
    if (tk_mi355x_llm_model_set_runner_slots(model, K) != TK_SUCCESS) return 6;
    job_t* jobs = (job_t*)calloc((size_t)K, sizeof *jobs);
    pthread_t* th = (pthread_t*)calloc((size_t)K, sizeof *th);
    tk_llm_config_t rc = {256, NULL, 0};
    for (int i = 0; i < K; ++i) {
        if (tk_llm_runner_create(&jobs[i].runner, model, &rc) != TK_SUCCESS) { fprintf(stderr, "runner %d: %s\n", i, tk_error_get_detail()); return 7; }
        for (int c = 0; c < 63; ++c) jobs[i].prompt[c] = (char)(97 + (i * 7 + c) % 26);
        jobs[i].prompt[63] = 0;
        jobs[i].n_tokens = N;
    }
    double dt = 0.0;
    uint64_t p0 = 0, r0 = 0, p1 = 0, r1 = 0;
    int32_t widest = 0;
    for (int rnd = 0; rnd < 2; ++rnd) { /* the first round captures the pass graphs */
        tk_mi355x_llm_model_batch_stats(model, &p0, &r0, &widest);
        const double t0 = now_s();
        for (int i = 0; i < K; ++i) pthread_create(&th[i], NULL, drive, &jobs[i]);
        for (int i = 0; i < K; ++i) pthread_join(th[i], NULL);
        dt = now_s() - t0;
    }
    tk_mi355x_llm_model_batch_stats(model, &p1, &r1, &widest);
    long total = 0;
    for (int i = 0; i < K; ++i) { total += jobs[i].count; if (jobs[i].rc != 0) { fprintf(stderr, "runner %d failed: %d\n", i, jobs[i].rc); return 8; } }
    printf("{\"runners\": %d, \"host_threads\": %d, \"driver\": \"C host (pthreads)\", \"tokens_per_runner\": %ld, \"wall_s\": %.4f, \"cycles_per_s_llm_only\": %.3f, "
           "\"tok_per_s\": %.1f, \"passes\": %llu, \"rows\": %llu, \"rows_per_pass\": %.2f, \"widest_pass\": %d}\n",
           K, K, total / K, dt, K / dt, (double)total / dt, (unsigned long long)(p1 - p0), (unsigned long long)(r1 - r0),
           (double)(r1 - r0) / (double)((p1 - p0) ? (p1 - p0) : 1), (int)widest);
    for (int i = 0; i < K; ++i) tk_llm_runner_destroy(&jobs[i].runner);
    if (tk_model_loader_unload_model(loader, &model) != TK_SUCCESS) return 9;
    tk_model_loader_destroy(&loader);
    free(jobs); free(th);
    return 0;
}
