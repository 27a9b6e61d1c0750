/*
 * tk_llm_layout.h — HBM layouts of the MI355X LLM path.
 *
 * GGUF k-quant blocks are kept bit-for-bit (same 144 B / 210 B per 256 weights, same
 * quantised values) but re-tiled at load time so that one wavefront's 16-byte-per-lane
 * load is a contiguous 1 KiB run that already IS an MFMA operand:
 *
 *  Weight tile = 16 weight rows x 256 k (one super-block column).  Lane l = (n = l & 15, g = l >> 4)
 *  owns weight row n and the 8-wide k-slice g of every 32-wide sub-block, i.e. exactly the
 *  B-operand fragment of v_mfma_i32_16x16x32_i8  (B[k = 8g + t][j = n]).
 *
 *  Q4_K tile (2304 B = 16 x 144):
 *      [0    ,1024)  load 0: lane l -> 4 dwords, dword s = sub-block s      (k = 32 s + 8 g + 0..7)
 *      [1024 ,2048)  load 1: lane l -> 4 dwords, dword s = sub-block 4 + s
 *                    dword byte t = q[k0 + t] | q[k0 + 4 + t] << 4   (k0 = 32 j + 8 g)
 *      [2048 ,2304)  16 rows x {f16 d, f16 dmin, 12 B packed 6-bit scales/mins} verbatim
 *  Q6_K tile (3360 B = 16 x 210): weights stored as 6-bit two's complement q' = (q - 32) & 63
 *      [0    ,2048)  two loads as above holding the LOW nibbles of q'
 *      [2048 ,3072)  lane l -> 4 dwords; dword u covers sub-blocks 2u, 2u+1: the 2 high bits of
 *                    the 4 weights that land in byte y of operand dword T_t sit at bits 8y+2t, 8y+2t+1
 *                    (T_0/T_1 = lo/hi dword of sub-block 2u, T_2/T_3 = lo/hi dword of 2u+1)
 *      [3072 ,3328)  16 rows x 16 int8 group scales verbatim
 *      [3328 ,3360)  16 rows x f16 d
 *  The kernel rebuilds int8 = q' << 2 = 4 (q - 32) with two shift/mask ops per dword and folds
 *  the factor 4 into the block scale (exact: power of two).
 *
 *  Tiles of one 16-row group are contiguous over k:  tiles[row_tile][block].
 *
 *  Activations (Q8_K-style, the reference CPU engine's numerics — oracle/tk_oracle_llm.cpp):
 *      aq  : int8   [K/64][4 g][16 row slot][2 sub-blocks][8]   == the A-operand image of v_mfma_i32_16x16x64_i8: lane
 *                    (slot, g) reads its 16 bytes (k-slice g of two consecutive 32-wide sub-blocks) with ONE ds_read_b128
 *                    (1.6x the LDS bandwidth of two 8-byte reads on gfx950, tools/lds_rate.hip); a K-range is one
 *                    contiguous run that is DMA'd into LDS
 *      ad  : float  [K/256][16]        block scale amax/127
 *      abs : int8   [K/256][2][16][8]  per-sub-block sums of the int8 values (the Q4_K "min" term) split as
 *                                      sum = 64 h + l: image 0 holds l (0..63), image 1 holds h, byte j = sub-block j —
 *                                      the A operand of the MFMA that contracts them with the 6-bit mins
 *      abs16: f16   [K/256][2][16][8]  the same sums as sum = 2 hh + ll (|hh| <= 2032, ll in {0, 1}: exact in f16) — the batched kernel
 *                                      contracts them with (2 m_j, m_j) in ONE v_mfma_f32_16x16x32_f16 (exact: every partial sum < 2^24)
 *  "row slot" b < 16 is a (sequence, position) row of the current pass; a pass holds up to TK_MAX_TILES such
 *  16-row M-tiles (row r lives in tile r / 16, slot r % 16), each with its own aq / ad / abs image.
 */
#ifndef TK_LLM_LAYOUT_H
#define TK_LLM_LAYOUT_H

#include <stdint.h>

#define TK_TILE_ROWS 16
#define TK_Q4K_TILE_BYTES 2304
#define TK_Q6K_TILE_BYTES 3360
#define TK_ROW_SLOTS 16  /* rows of one MFMA M-tile */
#define TK_MAX_TILES 16   /* M-tiles per pass: a weight tile is unpacked once and multiplied against all of them */
#define TK_MAX_ROWS (TK_ROW_SLOTS * TK_MAX_TILES)

/* per M-tile sizes; tile m of a buffer starts at m * (these) */
#define TK_AQ_BYTES(K) ((size_t)(K) * TK_ROW_SLOTS)
#define TK_AD_FLOATS(K) ((size_t)(K) / 256 * TK_ROW_SLOTS)
#define TK_ABS_BYTES(K) ((size_t)(K) / 256 * 256)

#endif
