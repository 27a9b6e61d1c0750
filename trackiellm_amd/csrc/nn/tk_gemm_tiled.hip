/* tk_gemm_tiled.hip — see tk_gemm_tiled.h */
#include "tk_gemm_tiled.h"

#include <atomic>
#include <mutex>

#include "../common/tk_exact_math.h"

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

#define TK_TW_SLOT_BYTES 65536

__device__ __forceinline__ uint4 tw_ldg_nt(const uint8_t* p) {
    const v4u v = __builtin_nontemporal_load((const v4u*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float tw_f16(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (uint16_t)h); }

/* s_waitcnt vmcnt(n) alone: until all but this wave's n youngest vector-memory operations are done (LDS-DMA pieces and weight requests
 * count together, in issue order) */
__device__ __forceinline__ void tw_wait_vmcnt(int n) {
#define TK_VMW(k) case k: __builtin_amdgcn_s_waitcnt(0x0F70 | ((k) & 15) | (((k) >> 4) << 14)); break;
    switch (n) {
        TK_VMW(0) TK_VMW(1) TK_VMW(2) TK_VMW(3) TK_VMW(4) TK_VMW(5) TK_VMW(6) TK_VMW(7) TK_VMW(8)
        default: __builtin_amdgcn_s_waitcnt(0x0F70); break;
    }
#undef TK_VMW
}

__device__ __forceinline__ float tw_act(float v, int act) {
    switch (act) {
        case TK_ACT_SILU: return tk_siluf(v);
        case TK_ACT_GELU: return tk_geluf(v);
        case TK_ACT_SIGMOID: return tk_sigmoidf(v);
        default: return v;
    }
}

/* WB = bytes per weight value (2: f16, 4: f32); MT = M-tiles per workgroup.  (Twelve weight pieces in flight for the Whisper decoder's
 * K = 384 launches — one round trip to the weights instead of three — measured 20 % SLOWER per launch: the whole activation image must
 * then land before the first MFMA, profiles/r04_perception.txt.) */
template <int MT, int WB, int PF /* weight pieces in flight per wave; a ring slot holds a multiple of PF chunks (the launcher's rk), so piece c sits in register set c % PF */>
__global__ __launch_bounds__(512) void k_gemm_tiled(TkTiledGemm a, int groups, int total_row_tiles, int rk /* k per ring slot */, int slot_bytes /* ring slot pitch */) {
    constexpr int LPC = WB == 4 ? 2 : 1; /* 16-byte requests per piece and lane */
    constexpr int PIECE = 512 * WB;      /* bytes of one (row tile, 32 k) piece */
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const int Kr = a.K / a.ks;
    const int ksi = blockIdx.x % a.ks;
    const int k0 = ksi * Kr;
    const int nslots = Kr / rk, cps = rk / 32, gps = rk / 16;
    /* grid y: the block of MT M-tiles this workgroup covers */
    const int row0 = blockIdx.y * MT * TK_TW_ROWS_PER_TILE;
    const float* a_img = a.a_img + (size_t)blockIdx.y * MT * a.a_ts;
    const int rows_here = a.nrows - row0 < MT * TK_TW_ROWS_PER_TILE ? a.nrows - row0 : MT * TK_TW_ROWS_PER_TILE;

    int rt = blockIdx.x / a.ks + wave * groups;
    const bool active = rt < total_row_tiles;
    if (!active) rt = 0;
    int seg = 0, col_base = 0;
    while (seg < a.nseg - 1 && rt >= a.row_tiles[seg]) { rt -= a.row_tiles[seg]; col_base += a.row_tiles[seg] * TK_TW_ROWS_PER_TILE; ++seg; }
    const int nchunk_total = a.K / 32, nchunk = Kr / 32;
    const uint8_t* wbase = a.tiles[seg] + ((size_t)rt * nchunk_total + k0 / 32) * PIECE + lane * (8 * WB);

    const int ppslot = gps * MT;
    const uint32_t voff = lane * 16;
    auto stage = [&](int s, int slot) { /* slot s of the range: [16-k group j][M-tile m] pieces of 1 KiB, piece p = j * MT + m dealt to wave p % nw */
        for (int p = wave; p < ppslot; p += nw) {
            const int j = p / MT, m = p % MT;
            const uint8_t* src = (const uint8_t*)(a_img + (size_t)m * a.a_ts + ((size_t)(k0 + s * rk) / 16 + j) * 256);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + voff),
                                             (__attribute__((address_space(3))) void*)(lds + (size_t)slot * slot_bytes + (size_t)p * 1024), 16, 0, 0);
        }
    };

    float acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][r] = 0.0f;

    /* a request past the range re-reads the last piece: no branch around a load, static wait counts */
    uint4 wq[PF][LPC];
    auto wload = [&](int c, uint4* dst) {
        const uint8_t* p = wbase + (size_t)(c < nchunk ? c : nchunk - 1) * PIECE;
        dst[0] = tw_ldg_nt(p);
        if (LPC == 2) dst[LPC - 1] = tw_ldg_nt(p + 16);
    };
    stage(0, 0);
#pragma unroll
    for (int i = 0; i < PF; ++i) wload(i, wq[i]);

#pragma unroll 1
    for (int s = 0; s < nslots; ++s) {
        /* slot s has landed once all but the PF * LPC youngest requests are done: those are weight requests issued after its DMA
         * (cps >= PF chunks follow every slot's DMA, the PF initial ones the first) */
        tw_wait_vmcnt(PF * LPC);
        __syncthreads();
        if (s + 1 < nslots) stage(s + 1, (s + 1) & 1);
        const uint8_t* slot = lds + (size_t)(s & 1) * slot_bytes + lane * 16;
#pragma unroll 1
        for (int cc = 0; cc < cps; cc += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int c = s * cps + cc + u;
                float wf[8];
                if (WB == 2) {
                    const uint4 w = wq[u][0];
                    wf[0] = tw_f16(w.x & 0xffffu); wf[1] = tw_f16(w.x >> 16); wf[2] = tw_f16(w.y & 0xffffu); wf[3] = tw_f16(w.y >> 16);
                    wf[4] = tw_f16(w.z & 0xffffu); wf[5] = tw_f16(w.z >> 16); wf[6] = tw_f16(w.w & 0xffffu); wf[7] = tw_f16(w.w >> 16);
                } else {
                    const uint4 w0 = wq[u][0], w1 = wq[u][LPC - 1];
                    wf[0] = __uint_as_float(w0.x); wf[1] = __uint_as_float(w0.y); wf[2] = __uint_as_float(w0.z); wf[3] = __uint_as_float(w0.w);
                    wf[4] = __uint_as_float(w1.x); wf[5] = __uint_as_float(w1.y); wf[6] = __uint_as_float(w1.z); wf[7] = __uint_as_float(w1.w);
                }
                wload(c + PF, wq[u]);
                const uint8_t* ap = slot + (size_t)(2 * (cc + u)) * MT * 1024;
                if (MT == 1) {
                    const v4f a0 = *(const v4f*)ap, a1 = *(const v4f*)(ap + 1024);
                    if (active) {
                        v4f d = {acc[0][0], acc[0][1], acc[0][2], acc[0][3]};
#pragma unroll
                        for (int t = 0; t < 4; ++t) d = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t], wf[t], d, 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < 4; ++t) d = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t], wf[4 + t], d, 0, 0, 0);
                        acc[0][0] = d[0]; acc[0][1] = d[1]; acc[0][2] = d[2]; acc[0][3] = d[3];
                    }
                } else {
                    /* M-tiles in pairs: the two accumulators' chains alternate (a chain step's 40-cycle dependent latency hides behind the
                     * other chain's MFMA) and the next pair's operand reads are issued before this pair's MFMAs */
                    v4f A0[2][2], A1[2][2]; /* [buffer][tile of the pair] */
                    A0[0][0] = *(const v4f*)(ap); A1[0][0] = *(const v4f*)(ap + MT * 1024);
                    A0[0][1] = *(const v4f*)(ap + 1024); A1[0][1] = *(const v4f*)(ap + (MT + 1) * 1024);
#pragma unroll
                    for (int m = 0; m < MT; m += 2) {
                        const int cur = (m >> 1) & 1, nxt = cur ^ 1;
                        if (m + 2 < MT) {
                            A0[nxt][0] = *(const v4f*)(ap + (m + 2) * 1024); A1[nxt][0] = *(const v4f*)(ap + (MT + m + 2) * 1024);
                            A0[nxt][1] = *(const v4f*)(ap + (m + 3) * 1024); A1[nxt][1] = *(const v4f*)(ap + (MT + m + 3) * 1024);
                        }
                        if (active) {
                            v4f d0 = {acc[m][0], acc[m][1], acc[m][2], acc[m][3]}, d1 = {acc[m + 1][0], acc[m + 1][1], acc[m + 1][2], acc[m + 1][3]};
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[cur][0][t], wf[t], d0, 0, 0, 0);
                                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[cur][1][t], wf[t], d1, 0, 0, 0);
                            }
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[cur][0][t], wf[4 + t], d0, 0, 0, 0);
                                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[cur][1][t], wf[4 + t], d1, 0, 0, 0);
                            }
                            acc[m][0] = d0[0]; acc[m][1] = d0[1]; acc[m][2] = d0[2]; acc[m][3] = d0[3];
                            acc[m + 1][0] = d1[0]; acc[m + 1][1] = d1[1]; acc[m + 1][2] = d1[2]; acc[m + 1][3] = d1[3];
                        }
                    }
                }
            }
        }
    }
    if (!active) return;
    const int g = lane >> 4;
    const int n = col_base + rt * TK_TW_ROWS_PER_TILE + (lane & 15);
    if (n >= a.n_valid) return;
    if (a.ks > 1) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m * TK_TW_ROWS_PER_TILE + 4 * g + r;
                if (row < rows_here) __builtin_nontemporal_store(acc[m][r], &a.out[((size_t)ksi * a.slab_rows + row0 + row) * a.ldc + n]);
            }
        return;
    }
    if (a.per_seg) { /* the segment is a layer of its own: column inside it, its bias, its destination */
        const int ns = rt * TK_TW_ROWS_PER_TILE + (lane & 15);
        if (ns >= a.seg_n[seg]) return;
        const float* bp = a.seg_bias[seg];
        const float bias = bp ? bp[ns] : 0.0f;
        float* outp = a.seg_out[seg];
        const int ldc = a.seg_ldc[seg];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m * TK_TW_ROWS_PER_TILE + 4 * g + r;
                if (row < rows_here) {
                    float v = acc[m][r];
                    if (bp || a.add_zero_bias) v = v + bias;
                    outp[(size_t)(row0 + row) * ldc + ns] = tw_act(v, a.act);
                }
            }
        return;
    }
    const float bias = a.bias ? a.bias[n] : 0.0f;
    /* operand-image position of column n as k of the next layer: k = 16 j + 4 t + g2 -> [M-tile][j][g2][row][t] (tk_gemm_tiled.h) */
    const size_t img_col = (size_t)(n >> 4) * 256 + (size_t)(n & 3) * 64 + ((n & 15) >> 2);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m * TK_TW_ROWS_PER_TILE + 4 * g + r;
            if (row < rows_here) {
                float v = acc[m][r];
                if (a.bias || a.add_zero_bias) v = v + bias;
                v = tw_act(v, a.act);
                if (a.residual) v = v + a.residual[(size_t)(row0 + row) * a.ldr + n];
                a.out[(size_t)(row0 + row) * a.ldc + n] = v;
                if (a.c_img) a.c_img[(size_t)((row0 + row) >> 4) * 16 * a.n_valid + img_col + (size_t)((row0 + row) & 15) * 4] = v;
            }
        }
}

/* ---- operand preparation ---- */

template <typename T>
__global__ void k_tile_weights(const T* src, int64_t N, int64_t K, T* tiles) {
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    const int64_t chunk = blockIdx.x, rt = blockIdx.y;
    const int64_t row = rt * 16 + n;
    T* dst = tiles + ((rt * (K / 32) + chunk) * 64 + lane) * 8;
#pragma unroll
    for (int t = 0; t < 8; ++t) dst[t] = row < N ? src[row * K + chunk * 32 + 4 * t + g] : (T)0;
}

size_t tk_tiled_weight_bytes(int64_t N, int64_t K, int wbytes) { return (size_t)((N + 15) / 16 * 16) * (size_t)K * (size_t)wbytes; }

void tk_launch_tile_weights(const void* src, int wbytes, int64_t N, int64_t K, uint8_t* tiles, hipStream_t s) {
    const dim3 grid((unsigned)(K / 32), (unsigned)((N + 15) / 16));
    if (wbytes == 2) hipLaunchKernelGGL((k_tile_weights<uint16_t>), grid, dim3(64), 0, s, (const uint16_t*)src, N, K, (uint16_t*)tiles);
    else hipLaunchKernelGGL((k_tile_weights<uint32_t>), grid, dim3(64), 0, s, (const uint32_t*)src, N, K, (uint32_t*)tiles);
}

/* one thread per image element (coalesced 16-byte stores); rows beyond `rows` read as zero */
__global__ void k_pack_a(const float* A, int64_t rows, int K, int lda, int round_f16, float* img) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; /* index of a float4 of the image */
    const int64_t mtiles = (rows + 15) / 16, per_tile = (int64_t)K * 4; /* float4 per M-tile = 16 K / 4 */
    if (i >= mtiles * per_tile) return;
    const int64_t mt = i / per_tile, q = i % per_tile;
    const int64_t j = q / 64;             /* 16-k group */
    const int g = (int)((q % 64) / 16), r = (int)(q % 16);
    const int64_t row = mt * 16 + r;
    v4f v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (row < rows) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float x = A[row * lda + 16 * j + 4 * t + g];
            v[t] = round_f16 ? tk_f16_to_f32(tk_f32_to_f16(x)) : x;
        }
    }
    *(v4f*)(img + i * 4) = v;
}

size_t tk_a_image_floats(int64_t rows, int64_t K) { return (size_t)((rows + 15) / 16 * 16) * (size_t)K; }

void tk_launch_pack_a(const float* A, int64_t rows, int K, int lda, int round_f16, float* img, hipStream_t s) {
    const int64_t n4 = (rows + 15) / 16 * (int64_t)K * 4;
    hipLaunchKernelGGL(k_pack_a, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, A, rows, K, lda, round_f16, img);
}

/* ---- launch ---- */

/* compute units of the calling thread's current device, cached PER DEVICE (the ABI takes a device ordinal per handle; launchers run with
 * the handle's device current and from several host threads) */
static int tk_tiled_num_cu() {
    static std::atomic<int> n[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int v = n[dev].load(std::memory_order_relaxed);
    if (v > 0) return v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
    n[dev].store(v, std::memory_order_relaxed);
    return v;
}

template <typename F>
static hipError_t tw_opt_in(F* fn) { return hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TK_TW_SLOT_BYTES); }

static std::atomic<bool> g_tw_opted[64];
static std::mutex g_tw_mu;
bool tk_gemm_tiled_prepare_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (g_tw_opted[dev].load(std::memory_order_acquire)) return true;
    std::lock_guard<std::mutex> lk(g_tw_mu);
    hipError_t e = hipSuccess;
#define TK_TW_OPT(MTV, PFV) do { if (e == hipSuccess) e = tw_opt_in(k_gemm_tiled<MTV, 2, PFV>); if (e == hipSuccess) e = tw_opt_in(k_gemm_tiled<MTV, 4, PFV>); } while (0)
    TK_TW_OPT(1, 4); TK_TW_OPT(2, 4); TK_TW_OPT(4, 4); TK_TW_OPT(8, 4); TK_TW_OPT(16, 2); TK_TW_OPT(8, 2); TK_TW_OPT(6, 2);
#undef TK_TW_OPT
    if (e != hipSuccess) return false;
    g_tw_opted[dev].store(true, std::memory_order_release);
    return true;
}

bool tk_launch_gemm_tiled(const TkTiledGemm& a, hipStream_t s) {
    if (a.nseg < 1 || a.nseg > 3 || a.ks < 1 || a.K <= 0 || a.K % (32 * a.ks) || a.nrows <= 0 || (a.wbytes != 2 && a.wbytes != 4) || (a.ks > 1 && a.nrows > TK_TW_MAX_BLOCK_ROWS)) return false;
    if ((a.per_seg || a.c_img) && a.ks != 1) return false;
    if (a.per_seg && (a.residual || a.c_img)) return false;
    if (a.c_img && (a.n_valid % 16 || a.nseg != 1)) return false;
    int row_tiles = 0;
    for (int i = 0; i < a.nseg; ++i) row_tiles += a.row_tiles[i];
    const int rows_blk = a.nrows < TK_TW_MAX_BLOCK_ROWS ? a.nrows : TK_TW_MAX_BLOCK_ROWS;
    int mt = rows_blk > 128 ? 16 : rows_blk > 64 ? 8 : rows_blk > 32 ? 4 : rows_blk > 16 ? 2 : 1;
    if (mt == 16 && a.nrows > TK_TW_MAX_BLOCK_ROWS && (a.K / a.ks) % 64 == 0) {
        /* many row blocks: two workgroups per CU (below), 512 at a time on the chip; of 128- and 96-row blocks the one whose last round
         * wastes less (N = 384 at 48 000 rows: 1 125 workgroups = 3 rounds of 8 M-tiles against 1 500 = 3 rounds of 6) */
        const long wg_per_blk = (row_tiles + 7) / 8;
        const long slots = 2L * tk_tiled_num_cu();
        const long c8 = (((a.nrows + 127) / 128) * wg_per_blk + slots - 1) / slots * 8;
        const long c6 = (((a.nrows + 95) / 96) * wg_per_blk + slots - 1) / slots * 6;
        mt = c6 < c8 ? 6 : 8;
    }
    const int ny = (a.nrows + mt * TK_TW_ROWS_PER_TILE - 1) / (mt * TK_TW_ROWS_PER_TILE);
    /* one pass of <= 256 rows (the LLM): spread the row tiles over the CUs, K-split ranges side by side; many row blocks: eight tiles
     * per workgroup share one activation ring */
    int groups, waves;
    if (ny == 1) {
        groups = 256 / a.ks;
        if (groups < 1) groups = 1;
        if (groups > row_tiles) groups = row_tiles;
        waves = (row_tiles + groups - 1) / groups;
        while (waves > 8) { groups *= 2; waves = (row_tiles + groups - 1) / groups; }
        if (waves == 1 && row_tiles >= 4) { /* fewer tiles than CUs: four one-tile waves (one per SIMD) share a workgroup's ring staging */
            waves = 4;
            groups = (row_tiles + 3) / 4;
        }
    } else {
        waves = row_tiles < 8 ? row_tiles : 8;
        groups = (row_tiles + waves - 1) / waves;
    }
    const int Kr = a.K / a.ks;
    /* k per ring slot: the largest multiple of 32 PF that divides the K range and fits a 64 KiB slot (rk / 16 * mt KiB) — a whole short
     * range in one slot spares a small launch its ring hand-overs (Whisper decoder: K = 384 at two M-tiles is one 48 KiB slot).
     * Many row blocks of 128 rows (the Whisper encoder's linears, 48000 rows): 32 KiB slots and two weight pieces in flight instead, so
     * that TWO workgroups share a CU and one's ring fill and epilogue run under the other's MFMAs. */
    const bool twin = ny > 1 && (mt == 8 || mt == 6) && Kr % 64 == 0;
    const int pf = (mt == 16 || twin) ? 2 : 4;
    const int slot_kib = twin ? 32 : 64;
    int rk = 0;
    for (int c = (slot_kib * 16 / mt) / (32 * pf) * (32 * pf); c >= 32 * pf; c -= 32 * pf)
        if (Kr % c == 0) { rk = c; break; }
    if (rk == 0) return false;
    if (Kr % rk) return false;
    const int slot_bytes = twin ? slot_kib * 1024 : TK_TW_SLOT_BYTES;
    const size_t ldsb = (size_t)2 * slot_bytes;
#define TK_TW_LAUNCH(MTV, WBV, PFV) hipLaunchKernelGGL((k_gemm_tiled<MTV, WBV, PFV>), dim3(groups * a.ks, ny), dim3(64 * waves), ldsb, s, a, groups, row_tiles, rk, slot_bytes)
#define TK_TW_WB(MTV, PFV) do { if (a.wbytes == 2) TK_TW_LAUNCH(MTV, 2, PFV); else TK_TW_LAUNCH(MTV, 4, PFV); } while (0)
    switch (mt) {
        case 1: TK_TW_WB(1, 4); break;
        case 2: TK_TW_WB(2, 4); break;
        case 4: TK_TW_WB(4, 4); break;
        case 6: TK_TW_WB(6, 2); break;
        case 8: if (twin) TK_TW_WB(8, 2); else TK_TW_WB(8, 4); break;
        default: TK_TW_WB(16, 2); break;
    }
#undef TK_TW_WB
#undef TK_TW_LAUNCH
    return true;
}
