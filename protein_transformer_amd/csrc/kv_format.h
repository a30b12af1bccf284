// The pre-split K / V format of the f16x2 attention kernels ("kv planes"): the key and value projections of every token as the
// two f16 terms the attention kernels multiply with, written ONCE by the epilogue of the QKV product (gemm_hp.hip) instead of
// being scaled, split and stored to LDS by every workgroup that streams them (attention_f16x2.hip: the forward kernel staged
// every 32-key tile of K and V of a (protein, head) once per 256 queries - a third of its VALU instructions).
//
// Reference: Attention.py:49-55 (one projection, x W_qkv^T + b, feeds Q, K and V of the scaled dot product) - the layout is
// this side's; the VALUES are those of the fp32 projection, split with the arithmetic of split_bf16.h (split_quad_f16):
//
//   x[row][d] * s = hi + lo + e,  |e| <= 2^-22 |x s|,   s = 2^e common to a GROUP OF FOUR consecutive tokens (all 64 d of
//   a head), max |x| s in [2^14, 2^15) - exactly the scaling groups the attention kernels formed themselves (Stage::store),
//   so the forward pass is bit-identical to the one that reads fp32 K / V.
//
// Memory: tokens in global tiles of 32 (token >> 5; T = B L tokens, L a multiple of 32 so that a tile never straddles two
// proteins).  Tile (which in {K = 0, V = 1}, head h, tile gt) is 8 KiB at byte ((which H + h) NT + gt) * 8192, NT = T / 32:
// [plane (hi, lo)][row 0..31][64 f16 = 8 chunks of 16 bytes], chunk c of row r stored at position c ^ rev3((r >> 1) & 7) -
// byte for byte the LDS image of a tile, so a stage is eight `global_load_lds_dwordx4` wave instructions (no VGPRs, no VALU,
// no ds_write), and conflict-free for the three ways the kernels read it:
//   * row fragments (ds_read_b128, lane l: row l & 31, chunk 2 step + (l >> 5)): the 16 lanes of a service group are rows
//     r0 .. r0 + 15 at one chunk c - their 16-byte slots ((r & 1) * 8 + (c ^ f(r))) of the 256-byte bank row are distinct
//     because f is a bijection of (r >> 1) & 7;
//   * transposing reads of 4 rows x 32 columns per 32 lanes (ds_read_tr16_b64; V in the forward kernel, Q / dO style tiles):
//     rows r0 .. r0 + 3 differ in bit 1, which f maps to chunk bit 2 - the two 64-byte halves of a 128-byte row;
//   * transposing reads of 8 rows x 16 columns per 32 lanes (the dQ pieces of the fused backward kernel): bits 1, 2 of the
//     row go to chunk bits 2, 1.
// The inverse group scales: float inv[((which H + h) NT + gt) * 8 + slot], slot = (g & 1) * 4 + (g >> 1) for group g = row >> 2 of
// the tile (a lane half of a 32 x 32 accumulator reads ITS four groups, g = 2 j + half, as one float4).
#pragma once
#include "common.h"

namespace ptkv {

constexpr int TILE_ROWS = 32, TILE_BYTES = 8192, PLANE_BYTES = 4096, ROW_BYTES = 128;

__host__ __device__ inline int rev3(int x) { return ((x & 1) << 2) | (x & 2) | ((x >> 2) & 1); }
// position (0..7) of chunk c (8 f16 = 16 bytes) in row r of a tile
__host__ __device__ inline int chunk_pos(int r, int c) { return c ^ rev3((r >> 1) & 7); }
// byte offset of (row r, first f16 d of a chunk-aligned or 4-aligned access) inside a plane
__host__ __device__ inline int plane_offset(int r, int d) { return r * ROW_BYTES + chunk_pos(r, d >> 3) * 16 + (d & 7) * 2; }
__host__ __device__ inline size_t tile_index(int which, int h, int gt, int H, int NT) { return ((size_t)which * H + h) * NT + gt; }
__host__ __device__ inline int group_slot(int g) { return (g & 1) * 4 + (g >> 1); }
__host__ __device__ inline size_t planes_bytes(int T, int H) { return (size_t)2 * H * ((T + 31) / 32) * TILE_BYTES; }
__host__ __device__ inline size_t inv_floats(int T, int H) { return (size_t)2 * H * ((T + 31) / 32) * 8; }

}  // namespace ptkv
