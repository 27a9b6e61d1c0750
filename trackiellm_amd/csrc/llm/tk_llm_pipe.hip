/*
 * tk_llm_pipe.hip — in-library stage hand-off of the layer-sharded LLM (protocol and rationale: tk_llm_pipe.h).
 */
#include "tk_llm_pipe.h"

#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include <mutex>
#include <vector>

#include "../common/tk_exact_math.h"

#define PQ(expr)                                                                                                   \
    do {                                                                                                           \
        hipError_t e__ = (expr);                                                                                   \
        if (e__ != hipSuccess) {                                                                                   \
            char b__[256];                                                                                         \
            snprintf(b__, sizeof b__, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            error = b__;                                                                                           \
            return false;                                                                                          \
        }                                                                                                          \
    } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

/* ---- device side ---------------------------------------------------------------------------------------------------------------- */

__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

/* spin until *word >= want; bounded by the pipe's timeout (st->timeout_ticks of the 100 MHz realtime counter).  Returns false on timeout
 * — and at once when the pipe has failed already: after the first timeout every wait still enqueued (the rest of a prompt, the decode
 * loop's graph replays) drains in microseconds instead of spinning its own full timeout. */
__device__ bool spin_until(const unsigned long long* word, unsigned long long want, const TkPipeState* st) {
    if (st->status != 0) return false;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (ld_sys(word) < want) {
        __builtin_amdgcn_s_sleep(8);
        if (__builtin_amdgcn_s_memrealtime() - t0 > st->timeout_ticks) return false;
    }
    return true;
}

/* consumer, first kernel of a pass: the next x message (sequence number recv_x + 1) has been published into my mailbox */
__global__ void k_pipe_wait_x(TkPipeBlock* mine, TkPipeState* st) {
    const unsigned long long seq = st->recv_x + 1;
    if (!spin_until(&mine->x_flag[seq % TK_PIPE_SLOTS][0], seq, st)) st->status = 1;
}

/* consumer: slot -> the session's residual stream (fp32 as it is, f16 widened); the last workgroup returns the credit to the producer
 * and advances recv_x.  A failed pipe (status set by this pass's wait or any earlier one) moves nothing and keeps its counters: the
 * protocol state stays what it was when the peer went missing. */
__global__ __launch_bounds__(256) void k_pipe_take_x(const TkPipeBlock* mine, TkPipeBlock* prev, TkPipeState* st, float* __restrict__ x, int n4 /* float4 groups */,
                                                      int slot_floats, int f16) {
    if (st->status != 0) return; /* uniform over the grid: written by an earlier kernel of this stream */
    const unsigned long long seq = st->recv_x + 1;
    const uint8_t* base = (const uint8_t*)mine + sizeof(TkPipeBlock) + (size_t)(seq % TK_PIPE_SLOTS) * slot_floats * 4;
    const int i = blockIdx.x * 256 + threadIdx.x;
    /* k_pipe_wait_x's acquire dropped stale lines of this slot only from the caches of the ONE workgroup that polled; this kernel's
     * workgroups sit behind other L2s (one per XCD) that may still hold the slot's contents of eight passes ago, and the slot was written
     * by a peer, not through them: acquire at system scope before the first payload read, whatever the launch boundary did
     * (tk_llm_pipe.h, "Coherence") */
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    if (i < n4) {
        if (f16) {
            const uint2 h = ((const uint2*)base)[i];
            v4f v;
            v[0] = tk_f16_to_f32((uint16_t)(h.x & 0xffffu)); v[1] = tk_f16_to_f32((uint16_t)(h.x >> 16));
            v[2] = tk_f16_to_f32((uint16_t)(h.y & 0xffffu)); v[3] = tk_f16_to_f32((uint16_t)(h.y >> 16));
            ((v4f*)x)[i] = v;
        } else {
            ((v4f*)x)[i] = ((const v4f*)base)[i];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int arrived = atomicAdd(&st->done_take, 1u);
        if (arrived == gridDim.x - 1) {
            st->done_take = 0;
            st->recv_x = seq;
            st_sys(&prev->credit_x[0], seq); /* the producer may reuse this slot once it wraps around */
        }
    }
}

/* producer, last kernel of a pass: x + (p_0 + ... + p_{ks-1}) — the pending residual update, slabs added in ascending order exactly as
 * k_rmsnorm_q8 / k_residual_fold do — stored straight into a slot of the NEXT stage's mailbox; the last workgroup to arrive publishes the
 * slot.  grid (d_model / 1024, nrows) */
__global__ __launch_bounds__(256) void k_pipe_send_x(const float* __restrict__ x, const float* __restrict__ partial, int ks, int n_total, int D, TkPipeBlock* next,
                                                      const TkPipeBlock* mine, TkPipeState* st, int slot_floats, int f16) {
    __shared__ int ok;
    const unsigned long long seq = st->sent_x + 1;
    if (threadIdx.x == 0) { /* credit: the consumer has freed the slot this message reuses */
        ok = st->status == 0;
        if (ok && seq > TK_PIPE_SLOTS && !spin_until(&mine->credit_x[0], seq - TK_PIPE_SLOTS, st)) { st->status = 1; ok = 0; }
    }
    __syncthreads();
    if (!ok) return; /* failed pipe: nothing is stored into a slot the consumer may still own, no counter moves */
    const int r = blockIdx.y, g = blockIdx.x * 256 + threadIdx.x;
    uint8_t* base = (uint8_t*)next + sizeof(TkPipeBlock) + (size_t)(seq % TK_PIPE_SLOTS) * slot_floats * 4;
    if (4 * g < D) {
        v4f v = *(const v4f*)(x + (int64_t)r * D + 4 * g);
        if (partial) {
            v4f o = *(const v4f*)(partial + (int64_t)r * n_total + 4 * g);
            for (int s = 1; s < ks; ++s) o = o + *(const v4f*)(partial + ((int64_t)s * TK_MAX_ROWS + r) * n_total + 4 * g);
            v = v + o;
        }
        const int64_t at = (int64_t)r * (D / 4) + g;
        if (f16) {
            uint2 h;
            h.x = (uint32_t)tk_f32_to_f16(v[0]) | ((uint32_t)tk_f32_to_f16(v[1]) << 16);
            h.y = (uint32_t)tk_f32_to_f16(v[2]) | ((uint32_t)tk_f32_to_f16(v[3]) << 16);
            ((uint2*)base)[at] = h;
        } else {
            ((v4f*)base)[at] = v;
        }
    }
    __threadfence_system(); /* this thread's payload stores are visible system-wide before the arrival below */
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int arrived = atomicAdd(&st->done_x, 1u);
        if (arrived == gridDim.x * gridDim.y - 1) {
            st->done_x = 0;
            st->sent_x = seq;
            __threadfence_system();
            st_sys(&next->x_flag[seq % TK_PIPE_SLOTS][0], seq);
        }
    }
}

/* last stage: the ids just sampled (d_tok) -> stage 0's id mailbox.  One workgroup. */
__global__ __launch_bounds__(256) void k_pipe_send_ids(const int32_t* __restrict__ tok, int nrows, TkPipeBlock* first, const TkPipeBlock* mine, TkPipeState* st) {
    __shared__ int ok;
    const unsigned long long seq = st->sent_ids + 1;
    if (threadIdx.x == 0) {
        ok = st->status == 0;
        if (ok && seq > TK_PIPE_SLOTS && !spin_until(&mine->credit_ids[0], seq - TK_PIPE_SLOTS, st)) { st->status = 1; ok = 0; }
    }
    __syncthreads();
    if (!ok) return;
    for (int i = threadIdx.x; i < nrows; i += 256) first->ids_payload[seq % TK_PIPE_SLOTS][i] = tok[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        st->sent_ids = seq;
        st_sys(&first->ids_flag[seq % TK_PIPE_SLOTS][0], seq);
    }
}

/* stage 0: wait for the ids of these rows' previous positions, make them the rows' tokens and note them in the session's history (what
 * stage 0 feeds at decode step i = what the last stage sampled at step i - 1).  One workgroup. */
__global__ __launch_bounds__(256) void k_pipe_take_ids(TkPipeBlock* mine, TkPipeBlock* last, TkPipeState* st, int32_t* __restrict__ tok, int nrows, int32_t* nsteps,
                                                        int32_t* hist, int hist_stride, int hist_cap) {
    __shared__ int ok;
    const unsigned long long seq = st->recv_ids + 1;
    if (threadIdx.x == 0) {
        ok = spin_until(&mine->ids_flag[seq % TK_PIPE_SLOTS][0], seq, st) ? 1 : 0;
        if (!ok) st->status = 1;
    }
    __syncthreads();
    if (!ok) return; /* failed pipe: the rows keep their tokens (the pass computes on them and its result is discarded with the error) */
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); /* the payload was written by a peer: every reading lane acquires, not only the poller */
    for (int i = threadIdx.x; i < nrows; i += 256) {
        const int32_t t = mine->ids_payload[seq % TK_PIPE_SLOTS][i];
        tok[i] = t;
        const int n = nsteps[i];
        if (n < hist_cap) { hist[(int64_t)n * hist_stride + i] = t; nsteps[i] = n + 1; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        st->recv_ids = seq;
        st_sys(&last->credit_ids[0], seq);
    }
}

/* stage 0: consume and discard `count` id messages.  A generation ends with one more id message sent (the last decode step's sample)
 * than stage 0 takes (it feeds the ids of steps 0 .. n - 1); when the host then describes new tokens itself, that message is stale and
 * must leave the FIFO before the next decode loop takes "the ids of the previous position" (the skew grew by one id per
 * generate() on one pipe).  One thread. */
__global__ void k_pipe_drain_ids(TkPipeBlock* mine, TkPipeBlock* last, TkPipeState* st, int count) {
    for (int c = 0; c < count; ++c) {
        const unsigned long long seq = st->recv_ids + 1;
        if (!spin_until(&mine->ids_flag[seq % TK_PIPE_SLOTS][0], seq, st)) { st->status = 1; return; }
        st->recv_ids = seq;
        st_sys(&last->credit_ids[0], seq);
    }
}

/* the rows' device-side positions move on by one (stages that do not sample: k_argmax does it on the last stage) */
__global__ void k_pipe_advance(int32_t* pos, int nrows) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nrows) pos[r] = pos[r] + 1;
}

/* stage 0, RCCL transport: the ids an ncclRecv has just put into `tok` are what the rows feed; note them in the history like k_pipe_take_ids */
__global__ void k_pipe_note_ids(const int32_t* __restrict__ tok, int nrows, int32_t* nsteps, int32_t* hist, int hist_stride, int hist_cap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    const int n = nsteps[i];
    if (n < hist_cap) { hist[(int64_t)n * hist_stride + i] = tok[i]; nsteps[i] = n + 1; }
}

/* ---- RCCL (loaded on first use: a host that never selects the collective transport does not need librccl) ------------------------ */
#include <dlfcn.h>
namespace {
typedef struct { char internal[128]; } tk_nccl_id;
struct RcclApi {
    int (*GetUniqueId)(tk_nccl_id*) = nullptr;
    int (*CommInitRank)(void**, int, tk_nccl_id, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string why;
};
enum { TK_NCCL_INT32 = 2, TK_NCCL_FLOAT32 = 7 }; /* ncclDataType_t (rccl.h: ncclInt32 = 2, ncclFloat32 = 7) */
}  // namespace
/* the prototypes and constants above are restated by hand so that the library does not LINK librccl; where the header is installed they
 * are checked against it at build time (a mismatch would corrupt data or hang silently at run time) */
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#include <type_traits>
static_assert(NCCL_UNIQUE_ID_BYTES == 128 && sizeof(ncclUniqueId) == sizeof(tk_nccl_id), "ncclUniqueId is not 128 opaque bytes");
static_assert((int)ncclInt32 == TK_NCCL_INT32 && (int)ncclFloat32 == TK_NCCL_FLOAT32, "ncclDataType_t values moved");
static_assert(sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int) && (int)ncclSuccess == 0, "nccl enums are not int-sized / ncclSuccess != 0");
static_assert(std::is_same<decltype(&ncclGetUniqueId), ncclResult_t (*)(ncclUniqueId*)>::value, "ncclGetUniqueId prototype");
static_assert(std::is_same<decltype(&ncclCommInitRank), ncclResult_t (*)(ncclComm_t*, int, ncclUniqueId, int)>::value, "ncclCommInitRank prototype");
static_assert(std::is_same<decltype(&ncclCommDestroy), ncclResult_t (*)(ncclComm_t)>::value, "ncclCommDestroy prototype");
static_assert(std::is_same<decltype(&ncclSend), ncclResult_t (*)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>::value, "ncclSend prototype");
static_assert(std::is_same<decltype(&ncclRecv), ncclResult_t (*)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>::value, "ncclRecv prototype");
static_assert(std::is_same<decltype(&ncclGetErrorString), const char* (*)(ncclResult_t)>::value, "ncclGetErrorString prototype");
static_assert(std::is_pointer<ncclComm_t>::value, "ncclComm_t is not a pointer");
#endif
namespace {
RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { api.why = std::string("librccl could not be loaded: ") + dlerror(); return; }
        api.GetUniqueId = (int (*)(tk_nccl_id*))dlsym(h, "ncclGetUniqueId");
        api.CommInitRank = (int (*)(void**, int, tk_nccl_id, int))dlsym(h, "ncclCommInitRank");
        api.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
        api.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclSend");
        api.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclRecv");
        api.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.Send && api.Recv && api.GetErrorString;
        if (!api.ok) api.why = "librccl lacks ncclSend / ncclRecv";
    });
    return api;
}
}  // namespace

bool tk_pipe_rccl_unique_id(unsigned char out[128], std::string* err) {
    RcclApi& r = rccl();
    if (!r.ok) { if (err) *err = r.why; return false; }
    tk_nccl_id id;
    const int rc = r.GetUniqueId(&id);
    if (rc != 0) { if (err) *err = std::string("ncclGetUniqueId: ") + r.GetErrorString(rc); return false; }
    memcpy(out, id.internal, 128);
    return true;
}

bool TkLlmPipe::connect_rccl(const unsigned char unique_id[128]) {
    if (n_stages == 1) return true;
    if (!unique_id) { error = "no RCCL unique id"; return false; }
    if (f16_) { error = "the RCCL transport carries the exact fp32 stream only"; return false; }
    int ndev = 0;
    PQ(hipGetDeviceCount(&ndev));
    if (ndev < 2) {
        error = "the RCCL transport needs one GPU per stage: " + std::to_string(ndev) + " device(s) visible to this process, " + std::to_string(n_stages) +
                " stages asked for (stages that share a GPU use the mailbox transport: tk_mi355x_pipe_connect / _connect_local)";
        return false;
    }
    RcclApi& r = rccl();
    if (!r.ok) { error = r.why; return false; }
    PQ(hipSetDevice(s_->model->device));
    tk_nccl_id id;
    memcpy(id.internal, unique_id, 128);
    const int rc = r.CommInitRank(&rccl_comm_, n_stages, id, stage);
    if (rc != 0) { rccl_comm_ = nullptr; error = std::string("ncclCommInitRank: ") + r.GetErrorString(rc) + " (two stages on one GPU?)"; return false; }
    PQ(hipMalloc((void**)&rccl_scratch_, TK_MAX_ROWS * sizeof(int32_t)));
    return true;
}

/* ---- host side ------------------------------------------------------------------------------------------------------------------ */

static std::mutex g_pipe_capture_mu;
static const char* const kFailedMsg =
    "a pipeline wait timed out: a neighbouring stage never published (peer process gone, or the stages walk different pass orders); the pipe "
    "is out of step with its neighbours and must be re-created";

TkLlmPipe::~TkLlmPipe() {
    if (s_ && s_->model) (void)hipSetDevice(s_->model->device);
    if (s_ && s_->stream) (void)hipStreamSynchronize(s_->stream);
    for (auto& gs : graph_) for (auto& g : gs) if (g) (void)hipGraphExecDestroy(g);
    if (next_ && next_ipc_) (void)hipIpcCloseMemHandle(next_);
    if (prev_ && prev_ipc_ && prev_ != next_) (void)hipIpcCloseMemHandle(prev_);
    if (rccl_comm_) (void)rccl().CommDestroy(rccl_comm_);
    if (rccl_scratch_) (void)hipFree(rccl_scratch_);
    if (mine_) (void)hipFree(mine_);
    if (st_) (void)hipFree(st_);
    if (h_rows_) (void)hipHostFree(h_rows_);
}

bool TkLlmPipe::init(TkLlmSession* session, int stg, int n, int layer0, int layer1, bool payload_f16, TkPipeHandle* out_handle) {
    if (!session || !session->model) { error = "no session"; return false; }
    const TkLlmHParams& h = session->model->hp;
    if (n < 1 || stg < 0 || stg >= n || layer0 < 0 || layer1 < layer0 || layer1 > h.n_layer) { error = "bad stage / layer range"; return false; }
    if (stg == n - 1 && layer1 != h.n_layer) { error = "the last stage must end at the last layer"; return false; }
    if (stg == 0 && layer0 != 0) { error = "stage 0 must start at layer 0"; return false; }
    if (h.d_model % 4) { error = "d_model must be a multiple of 4"; return false; }
    s_ = session; stage = stg; n_stages = n; l0 = layer0; l1 = layer1; f16_ = payload_f16;
    PQ(hipSetDevice(s_->model->device));
    block_bytes_ = sizeof(TkPipeBlock) + (size_t)TK_PIPE_SLOTS * TK_MAX_ROWS * h.d_model * 4;
    /* the mailbox is polled and written by OTHER agents while kernels of this one run: fine-grained device memory (tk_llm_pipe.h,
     * "Coherence").  TK_MI355X_PIPE_COARSE=1 keeps plain hipMalloc memory for an A/B on one device. */
    const char* coarse = getenv("TK_MI355X_PIPE_COARSE");
    fine_grained_ = !(coarse && coarse[0] == '1');
    if (fine_grained_) PQ(hipExtMallocWithFlags((void**)&mine_, block_bytes_, hipDeviceMallocFinegrained));
    else PQ(hipMalloc((void**)&mine_, block_bytes_));
    PQ(hipMemset(mine_, 0, sizeof(TkPipeBlock)));
    PQ(hipMalloc((void**)&st_, sizeof(TkPipeState)));
    TkPipeState st0{};
    double tmo = TK_PIPE_TIMEOUT_S;
    if (const char* e = getenv("TK_MI355X_PIPE_TIMEOUT_S")) { const double v = atof(e); if (v > 0.0 && v < 3600.0) tmo = v; }
    st0.timeout_ticks = (unsigned long long)(tmo * 1e8); /* s_memrealtime: 100 MHz */
    PQ(hipMemcpy(st_, &st0, sizeof st0, hipMemcpyHostToDevice));
    PQ(hipHostMalloc((void**)&h_rows_, (size_t)64 * 3 * TK_MAX_ROWS * sizeof(int32_t), hipHostMallocDefault));
    PQ(hipDeviceSynchronize());
    if (out_handle) {
        memset(out_handle, 0, sizeof *out_handle);
        hipIpcMemHandle_t ih;
        PQ(hipIpcGetMemHandle(&ih, mine_));
        static_assert(sizeof(ih) <= sizeof(out_handle->ipc), "hipIpcMemHandle_t grew");
        memcpy(out_handle->ipc, &ih, sizeof ih);
        out_handle->bytes = block_bytes_;
        out_handle->device = s_->model->device;
        out_handle->pid = (int32_t)getpid();
    }
    return true;
}

bool TkLlmPipe::connect(const TkPipeHandle* next, const TkPipeHandle* prev) {
    if (n_stages == 1) return true;
    if (!next || !prev) { error = "both neighbours' handles are needed"; return false; }
    if (next->bytes != block_bytes_ || prev->bytes != block_bytes_) { error = "a neighbour's mailbox has another geometry (model or build differ)"; return false; }
    PQ(hipSetDevice(s_->model->device));
    auto open = [&](const TkPipeHandle* hd, TkPipeBlock** out) -> bool {
        if (hd->pid == (int32_t)getpid()) { error = "a handle of this process: use connect_local for stages that share a process"; return false; }
        hipIpcMemHandle_t ih;
        memcpy(&ih, hd->ipc, sizeof ih);
        void* p = nullptr;
        PQ(hipIpcOpenMemHandle(&p, ih, hipIpcMemLazyEnablePeerAccess));
        *out = (TkPipeBlock*)p;
        return true;
    };
    if (!open(next, &next_)) return false;
    next_ipc_ = true;
    if (memcmp(next->ipc, prev->ipc, sizeof next->ipc) == 0) { prev_ = next_; prev_ipc_ = false; } /* two stages: one neighbour, mapped once */
    else { if (!open(prev, &prev_)) return false; prev_ipc_ = true; }
    return true;
}

bool TkLlmPipe::connect_local(TkLlmPipe* next, TkLlmPipe* prev) {
    if (n_stages == 1) return true;
    if (!next || !prev || !next->mine_ || !prev->mine_) { error = "both neighbours are needed"; return false; }
    if (next->block_bytes_ != block_bytes_ || prev->block_bytes_ != block_bytes_) { error = "a neighbour's mailbox has another geometry"; return false; }
    PQ(hipSetDevice(s_->model->device));
    for (TkLlmPipe* o : {next, prev}) {
        const int od = o->s_->model->device;
        if (od != s_->model->device) {
            int can = 0;
            PQ(hipDeviceCanAccessPeer(&can, s_->model->device, od));
            if (!can) { error = "no peer access between the two stages' GPUs"; return false; }
            const hipError_t e = hipDeviceEnablePeerAccess(od, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { error = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e); return false; }
            (void)hipGetLastError();
        }
    }
    next_ = next->mine_;
    prev_ = prev->mine_;
    return true;
}

/* one pass of this stage on the session's stream: [take the ids (stage 0) | wait for + take the stream] -> layers [l0, l1) -> [hand the
 * stream on | sample and return the ids].  take_ids: stage 0 feeds the ids of the id mailbox instead of what d_tok holds;
 * advance_pos: the rows' device-side positions move on by one afterwards (a sampling pass: k_argmax does it on the last stage, the
 * other stages follow suit here, so every stage's positions agree when a decode loop starts) */
void TkLlmPipe::enqueue_stage(int nrows, bool take_ids, bool head, bool advance_pos, bool fused_attn) {
    const TkLlmHParams& h = s_->model->hp;
    hipStream_t st = s_->stream;
    const int D = h.d_model, slot_floats = TK_MAX_ROWS * D;
    const bool first = stage == 0, last = stage == n_stages - 1;
    if (rccl_comm_) { /* the collective transport: one ncclRecv / ncclSend per boundary on this stage's stream, matched by the neighbour's */
        RcclApi& r = rccl();
        auto chk = [&](int rc, const char* what) { if (rc != 0 && s_->launch_error.empty()) s_->launch_error = std::string(what) + ": " + r.GetErrorString(rc); };
        if (first) {
            if (take_ids) {
                chk(r.Recv(s_->d_tok, (size_t)nrows, TK_NCCL_INT32, n_stages - 1, rccl_comm_, st), "ncclRecv (ids)");
                hipLaunchKernelGGL(k_pipe_note_ids, dim3((nrows + 63) / 64), dim3(64), 0, st, s_->d_tok, nrows, s_->d_nsteps, s_->d_hist, TK_MAX_ROWS, s_->hist_cap);
            }
        } else {
            chk(r.Recv(s_->x, (size_t)nrows * D, TK_NCCL_FLOAT32, stage - 1, rccl_comm_, st), "ncclRecv (stream)");
        }
        const bool sample = head && last;
        s_->enqueue_range(nrows, l0, l1, first, !last, sample, fused_attn); /* fold_out: the stream that leaves is complete */
        if (!last) chk(r.Send(s_->x, (size_t)nrows * D, TK_NCCL_FLOAT32, stage + 1, rccl_comm_, st), "ncclSend (stream)");
        else if (sample) chk(r.Send(s_->d_tok, (size_t)nrows, TK_NCCL_INT32, 0, rccl_comm_, st), "ncclSend (ids)");
        if (advance_pos && !sample) hipLaunchKernelGGL(k_pipe_advance, dim3((nrows + 63) / 64), dim3(64), 0, st, s_->d_pos, nrows);
        return;
    }
    if (first) {
        if (take_ids && n_stages > 1)
            hipLaunchKernelGGL(k_pipe_take_ids, dim3(1), dim3(256), 0, st, mine_, prev_, st_, s_->d_tok, nrows, s_->d_nsteps, s_->d_hist, TK_MAX_ROWS, s_->hist_cap);
    } else {
        hipLaunchKernelGGL(k_pipe_wait_x, dim3(1), dim3(1), 0, st, mine_, st_);
        const int n4 = nrows * D / 4;
        hipLaunchKernelGGL(k_pipe_take_x, dim3((n4 + 255) / 256), dim3(256), 0, st, mine_, prev_, st_, s_->x, n4, slot_floats, f16_ ? 1 : 0);
    }
    const bool sample = head && last;
    s_->enqueue_range(nrows, l0, l1, first, false, sample, fused_attn);
    if (!last) {
        const float* partial = l1 > l0 ? s_->partial : nullptr;
        hipLaunchKernelGGL(k_pipe_send_x, dim3((D / 4 + 255) / 256, nrows), dim3(256), 0, st, s_->x, partial, s_->last_ks_res, D, D, next_, mine_, st_, slot_floats,
                           f16_ ? 1 : 0);
    } else if (sample && n_stages > 1) {
        hipLaunchKernelGGL(k_pipe_send_ids, dim3(1), dim3(256), 0, st, s_->d_tok, nrows, next_, mine_, st_);
    }
    if (advance_pos && !sample) hipLaunchKernelGGL(k_pipe_advance, dim3((nrows + 63) / 64), dim3(64), 0, st, s_->d_pos, nrows);
}

bool TkLlmPipe::pass(int nrows, const int32_t* seq, const int32_t* pos, const int32_t* tok, bool head) {
    const TkLlmHParams& h = s_->model->hp;
    if (nrows <= 0 || nrows > TK_MAX_ROWS || !seq || !pos) { error = "nrows must be in [1, 256] and (seq, pos) given"; return false; }
    if (n_stages > 1 && !rccl_comm_ && (!next_ || !prev_)) { error = "the pipe is not connected"; return false; }
    if (failed_) { error = kFailedMsg; return false; }
    bool distinct = true;
    for (int r = 0; r < nrows; ++r) {
        if (seq[r] < 0 || seq[r] >= s_->max_seq || pos[r] < 0 || pos[r] >= s_->max_ctx || (stage == 0 && tok && (tok[r] < 0 || tok[r] >= h.vocab))) { error = "row out of range (sequence id, position or token id)"; return false; }
        for (int q = 0; q < r && distinct; ++q) distinct = seq[q] != seq[r];
    }
    PQ(hipSetDevice(s_->model->device));
    int32_t* hb = h_rows_ + (size_t)(h_next_ % 64) * 3 * TK_MAX_ROWS; /* pinned: the copies below run when the stream gets there */
    if (h_next_ >= 64) PQ(hipStreamSynchronize(s_->stream)); /* ring wrapped: rare (prompts of more than 64 passes) */
    h_next_ = h_next_ >= 64 ? 1 : h_next_ + 1;
    memcpy(hb, seq, nrows * 4);
    memcpy(hb + TK_MAX_ROWS, pos, nrows * 4);
    PQ(hipMemcpyAsync(s_->d_seq, hb, nrows * 4, hipMemcpyHostToDevice, s_->stream));
    PQ(hipMemcpyAsync(s_->d_pos, hb + TK_MAX_ROWS, nrows * 4, hipMemcpyHostToDevice, s_->stream));
    const bool from_tok = stage == 0 && tok != nullptr;
    if (from_tok) {
        memcpy(hb + 2 * TK_MAX_ROWS, tok, nrows * 4);
        PQ(hipMemcpyAsync(s_->d_tok, hb + 2 * TK_MAX_ROWS, nrows * 4, hipMemcpyHostToDevice, s_->stream));
    }
    if (s_->mask_rows_dirty) { PQ(hipMemsetAsync(s_->d_mask_row, 0xFF, TK_MAX_ROWS * 4, s_->stream)); s_->mask_rows_dirty = false; }
    s_->launch_error.clear();
    /* id FIFO bookkeeping (every stage enqueues the same schedule, so stage 0 can count the last stage's sends): host-given tokens make
     * whatever the FIFO still holds stale — the previous generation's last sample — so it is drained before this pass */
    const bool take = stage == 0 && !from_tok && n_stages > 1;
    if (stage == 0 && from_tok && n_stages > 1 && ids_outstanding_ > 0) {
        if (rccl_comm_) { /* every ncclSend of the last stage needs its ncclRecv: the stale messages are received into a scratch buffer */
            for (int rows : id_msg_rows_) {
                const int rc = rccl().Recv(rccl_scratch_, (size_t)rows, TK_NCCL_INT32, n_stages - 1, rccl_comm_, s_->stream);
                if (rc != 0) { error = std::string("ncclRecv (drain): ") + rccl().GetErrorString(rc); return false; }
            }
            id_msg_rows_.clear();
        } else {
            hipLaunchKernelGGL(k_pipe_drain_ids, dim3(1), dim3(1), 0, s_->stream, mine_, prev_, st_, ids_outstanding_);
        }
        ids_outstanding_ = 0;
    }
    /* the id FIFO has TK_PIPE_SLOTS slots and stage 0 only returns credits when it takes a message: more sampling passes than that without a
     * take in between would park the last stage on credits until the next generation's drain.  One row group per pipe (LibPipeline, bench.py)
     * keeps at most two outstanding; a host that interleaves many sampled prompts on one pipe is told so instead of being slowed silently. */
    /* the count is stage 0's: it is the only stage that takes or drains, so only there does the number mean anything (on the other stages it
     * would grow by one per generation and refuse the 8th prompt of a long-lived pipe) */
    if (stage == 0 && head && n_stages > 1 && !take && ids_outstanding_ >= TK_PIPE_SLOTS - 1) {
        error = "more than " + std::to_string(TK_PIPE_SLOTS - 1) + " sampled ids are waiting in the id mailbox: decode (or feed host tokens) before sampling further prompts on this pipe";
        return false;
    }
    if (take) { --ids_outstanding_; if (rccl_comm_ && !id_msg_rows_.empty()) id_msg_rows_.pop_front(); }
    if (stage == 0 && head && n_stages > 1) { ++ids_outstanding_; if (rccl_comm_) id_msg_rows_.push_back(nrows); }
    /* host-described passes go eagerly (their row tables differ); distinct rows take the fused-attention form, as forward() does */
    s_->choose_attention(pos, nrows);
    host_top_ = 0; /* where decode() continues from: the rows' positions advance on the device after a sampling pass */
    for (int r = 0; r < nrows; ++r) host_top_ = pos[r] > host_top_ ? pos[r] : host_top_;
    if (head) ++host_top_;
    enqueue_stage(nrows, take, head, head, distinct);
    if (!s_->launch_error.empty()) { error = s_->launch_error; return false; }
    PQ(hipGetLastError());
    return true;
}

bool TkLlmPipe::decode(int nrows, int n_steps) {
    if (nrows <= 0 || nrows > TK_MAX_ROWS || n_steps <= 0 || n_steps > s_->hist_cap) { error = "nrows must be in [1, 256] and n_steps within the session's context"; return false; }
    if (n_stages > 1 && !rccl_comm_ && (!next_ || !prev_)) { error = "the pipe is not connected"; return false; }
    if (failed_) { error = kFailedMsg; return false; }
    if (stage == 0 && n_stages > 1 && ids_outstanding_ < 1) { error = "decode() without a sampling pass before it: the id mailbox would be empty"; return false; }
    PQ(hipSetDevice(s_->model->device));
    if (s_->mask_rows_dirty) { PQ(hipMemsetAsync(s_->d_mask_row, 0xFF, TK_MAX_ROWS * 4, s_->stream)); s_->mask_rows_dirty = false; }
    PQ(hipMemsetAsync(s_->d_nsteps, 0, TK_MAX_ROWS * 4, s_->stream));
    const char* ng = getenv("TK_MI355X_NO_GRAPH");
    const bool use_graph = !(ng && ng[0] == '1') && !rccl_comm_; /* collective calls are launched eagerly */
    /* the attention form (fused / long-context three-launch form) follows the position the loop stands at, as TkLlmSession::decode does: the
     * host knows it from the last pass() and counts the steps; one graph per (form, row count) */
    auto capture = [&](hipGraphExec_t* slot) -> bool {
        if (*slot) return true;
        std::lock_guard<std::mutex> lk(g_pipe_capture_mu);
        hipGraph_t g = nullptr;
        PQ(hipStreamBeginCapture(s_->stream, hipStreamCaptureModeRelaxed));
        s_->launch_error.clear();
        enqueue_stage(nrows, true, true, true, true);
        const hipError_t e_end = hipStreamEndCapture(s_->stream, &g);
        hipError_t e_inst = hipSuccess;
        if (e_end == hipSuccess && s_->launch_error.empty()) e_inst = hipGraphInstantiate(slot, g, nullptr, nullptr, 0);
        if (g) (void)hipGraphDestroy(g);
        if (e_end != hipSuccess || e_inst != hipSuccess || !s_->launch_error.empty()) {
            *slot = nullptr;
            (void)hipGetLastError();
            error = !s_->launch_error.empty() ? s_->launch_error : std::string("graph capture of a pipeline pass failed: ") + hipGetErrorString(e_end != hipSuccess ? e_end : e_inst);
            return false;
        }
        return true;
    };
    /* a step takes one id message and (the last stage) sends one: the FIFO's length is unchanged, its row counts all become nrows */
    if (rccl_comm_ && stage == 0 && n_stages > 1) { id_msg_rows_.assign(id_msg_rows_.size(), nrows); }
    for (int i = 0; i < n_steps; ++i) {
        s_->choose_attention_top(host_top_ + i, nrows);
        if (use_graph) {
            hipGraphExec_t* slot = &graph_[s_->long_pass ? 1 : 0][nrows];
            if (!capture(slot)) return false;
            PQ(hipGraphLaunch(*slot, s_->stream));
        } else { s_->launch_error.clear(); enqueue_stage(nrows, true, true, true, true); if (!s_->launch_error.empty()) { error = s_->launch_error; return false; } }
    }
    host_top_ += n_steps;
    PQ(hipGetLastError());
    return true;
}

bool TkLlmPipe::sync(int32_t* out_tokens, int n_steps) {
    PQ(hipSetDevice(s_->model->device));
    if (out_tokens && n_steps > 0) {
        if (n_steps > s_->hist_cap) { error = "n_steps exceeds the session's history"; return false; }
        PQ(hipMemcpyAsync(out_tokens, s_->d_hist, (size_t)n_steps * TK_MAX_ROWS * 4, hipMemcpyDeviceToHost, s_->stream));
    }
    TkPipeState hs{};
    PQ(hipMemcpyAsync(&hs, st_, sizeof hs, hipMemcpyDeviceToHost, s_->stream));
    PQ(hipStreamSynchronize(s_->stream));
    h_next_ = 0;
    if (hs.status != 0 || failed_) {
        /* the status word stays set and so does failed_: this stage's sequence numbers no longer agree with its neighbours', the pipe has to
         * be re-created (all stages) — every further pass / decode / sync fails at once */
        failed_ = true;
        error = kFailedMsg;
        return false;
    }
    return true;
}
