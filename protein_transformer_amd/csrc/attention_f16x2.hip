// Fused masked multi-head self-attention on the f16 matrix pipe with two-term operand splitting (dk = 64 and 32).
//
// Same operator, interface, masking, dropout hash and outputs as attention.hip / attention_split.hip
//   /root/reference/protein_transformer/models/transformer/Attention.py:14-22,55-68
// but every f32 matrix product (QK^T, PV and the five products of the backward pass) is evaluated as THREE
// v_mfma_f32_32x32x16_f16 products of operands that were scaled by a power of two and split into two f16 terms
// (x s = h1 + h2 + e, |e| <= 2^-22 |x s|: the f16x2 arithmetic of gemm_f16x2.hip) - half the matrix-pipe time and
// about a third of the splitting work of the three-term bf16 kernels, with an error bound that is relative to the
// largest element of a scaling group instead of to every element.  Selected by arith = PTAMD_GEMM_AUTO / _F16X2.
//
// Scaling groups - no pass over the operands, no extra launch, no scale arrays in the interface:
//   * the operand whose rows live in lanes (Q and dO in the forward / dQ kernels, K and V in the dK/dV kernel) is scaled
//     by the power of two of ITS row maximum (a lane holds half a row, one exchange between the lane halves);
//   * a tile that streams through LDS is scaled per group of FOUR rows by the wavefront that stages them (64 lanes hold
//     exactly four rows of 64 floats), which publishes the inverse scale beside the tile.  In the 32 x 32 accumulator
//     layout a register quadruple (r >> 2) of a lane half is one such group, so
//       - where the tile rows are rows of the product (S^T = K Q^T, dP^T = V dO^T) the inverse scales are four
//         per-lane constants folded into the factors the soft-max needs anyway;
//       - where the tile rows are the CONTRACTED index (O^T += V^T P^T, dQ^T += K^T dS^T, dK^T += Q^T dS, dV^T += dO^T P)
//         the inverse group scale is folded into the register operand (P or dS) before it is split, together with a
//         common power of two that is adapted on line like the running maximum of the soft-max: when a later tile
//         needs a smaller one the accumulators are rescaled (rare: the factor has 4 bits of headroom).
// Decomposition, LDS format, dropout hash and prefetching are those of attention_split.hip (8 wavefronts = 256 queries
// or keys per workgroup, 32-row tiles, scores transposed so that a soft-max row is lane-local), with two planes per
// tile instead of three (48 KB of LDS instead of 72).
#include <stdlib.h>

#include "attn_dropout.h"
#include "kv_format.h"
#include "split_bf16.h"

namespace ptattn16 {
using namespace ptsplit;

// NW wavefronts per workgroup, 32 queries (or keys) each: 8 (256 per workgroup) or, when that leaves half of the CUs
// without a workgroup (few proteins), 4 - the same wavefronts then run one per SIMD instead of two, which is what
// bounds a workgroup (VALU issue), at the price of staging twice the share of every tile.
// PARTS = 2 (with NW = 4, when even those cover at most half of the CUs): the streamed dimension (keys in the forward /
// dQ kernels, queries in the dK/dV kernel) is cut in two halves, wavefronts 0-1 walk the first, 2-3 the second - half
// the serial tile loop per wavefront, which is what a launch of few workgroups is bound by - and the two partial
// results of a query (key) group are combined through LDS at the end: soft-max statistics and accumulator scales for
// the forward pass, plain sums (after the accumulator scales) for the gradients.  Fixed order: deterministic.
constexpr int TR = 32;           // rows (keys or queries) of an LDS tile
constexpr float LOG2E = 1.4426950408889634f;
constexpr float TWO14 = 16384.f, INV_TWO14 = 1.f / 16384.f;

__device__ __forceinline__ int crow(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }
__device__ __forceinline__ uint32_t abs_bits(float x) { return __float_as_uint(x) & 0x7fffffffu; }
// 1 / x for x = 0 (-> 2^127, finite) or a power of two in [2^-126, 2^127]
__device__ __forceinline__ float inv_pow2(float x) { return __uint_as_float((254u << 23) - __float_as_uint(x)); }
__device__ __forceinline__ uint32_t umax4(const float4 &v) {
  return max(max(abs_bits(v.x), abs_bits(v.y)), max(abs_bits(v.z), abs_bits(v.w)));
}

// maximum over groups of 32 or 64 adjacent lanes (in every lane of the group): four DPP steps inside the rows of 16,
// then one or two lane exchanges
template <int LANES>
__device__ __forceinline__ uint32_t group_umax(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));  // row_half_mirror
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));  // row_mirror
  v = max(v, (uint32_t)__shfl_xor((int)v, 16, 64));
  if (LANES == 64) v = max(v, (uint32_t)__shfl_xor((int)v, 32, 64));
  return v;
}

// x where bit r of `keep` is set, +0 elsewhere: a sign-extending 1-bit field extract (0 or ~0) and an AND - two
// instructions per element instead of bit test + compare + select
__device__ __forceinline__ float keep_or_zero(float x, uint32_t keep, int r) {
  return __uint_as_float(__float_as_uint(x) & (uint32_t)__builtin_amdgcn_sbfe((int)keep, r, 1));
}

__device__ __forceinline__ f32x16 mfma3(const f16x8 (&a)[2], const f16x8 (&b)[2], f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c, 0, 0, 0);
  return c;
}
// eight f32 -> the two f16x8 MFMA operands of (x[0..3] s0, x[4..7] s1)
__device__ __forceinline__ void split8g(const float (&x)[8], float s0, float s1, f16x8 (&f)[2]) {
  uint2 a1, a2, b1, b2;
  split_quad_f16(x[0], x[1], x[2], x[3], s0, s0, s0, s0, a1, a2);
  split_quad_f16(x[4], x[5], x[6], x[7], s1, s1, s1, s1, b1, b2);
  f[0] = __builtin_bit_cast(f16x8, (u32x4){a1.x, a1.y, b1.x, b1.y});
  f[1] = __builtin_bit_cast(f16x8, (u32x4){a2.x, a2.y, b2.x, b2.y});
}

// ---- LDS image of a [32][64] f32 tile as two f16 planes: row format and swizzle of split_bf16.h (Tile64)
constexpr int T_LD = 96;
struct Tile2 {
  static constexpr int PLANE = TR * T_LD;
  static constexpr int ELEMS = 2 * PLANE;
  static __device__ __forceinline__ int offset(int row, int d) {
    return row * T_LD + ((((d >> 3) ^ (row >> 2)) & 3) | ((d >> 3) & 4)) * 8 + (d & 7);
  }
  static __device__ __forceinline__ void store4(unsigned short *__restrict__ s, int row, int d, const float4 &v, float sc) {
    uint2 t1, t2;
    split_quad_f16(v.x, v.y, v.z, v.w, sc, sc, sc, sc, t1, t2);
    const int off = offset(row, d);
    *reinterpret_cast<uint2 *>(s + off) = t1;
    *reinterpret_cast<uint2 *>(s + PLANE + off) = t2;
  }
  // lane l holds tile row (l & 31), d = 16 step + 8 (l >> 5) + 0..7
  static __device__ __forceinline__ void frag_rows(const unsigned short *__restrict__ s, int step, int lane, f16x8 (&f)[2]) {
    const unsigned short *q = s + offset(lane & 31, 16 * step + 8 * (lane >> 5));
#pragma unroll
    for (int t = 0; t < 2; ++t) f[t] = *reinterpret_cast<const f16x8 *>(q + t * PLANE);
  }
  // lane l holds column d0 + (l & 31) of the tile rows kb + 4 (l >> 5) + {0..3} and kb + 8 + 4 (l >> 5) + {0..3}: the k
  // order in which a 32 x 32 accumulator holds its rows (split_bf16.h)
  static __device__ __forceinline__ void frag_cols(const unsigned short *__restrict__ s, int kb, int d0, int lane,
                                                   f16x8 (&f)[2]) {
    const int q16 = lane & 15;
    const int row = kb + 4 * (lane >> 5) + (q16 >> 2);
    const unsigned short *q0 = s + offset(row, d0 + (lane & 16) + 4 * (q16 & 3));
    const unsigned short *q1 = s + offset(row + 8, d0 + (lane & 16) + 4 * (q16 & 3));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(q0 + t * PLANE));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(q1 + t * PLANE));
      const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      f[t] = __builtin_bit_cast(f16x8, both);
    }
  }
};

// ---- the same tile UNPADDED (round 6): [plane][32 rows][64 f16] = 4 KiB per plane with the chunk swizzle of kv_format.h
// (conflict-free for row fragments and both transposing reads) instead of 96-element rows: 16 KB per {K, V} buffer pair instead
// of 24, so that FOUR key quarters of a workgroup fit LDS (attn_fwd_f16x2_kernel<64, 8, 4, Tile2U>: 128 KB)
struct Tile2U {
  static constexpr int PLANE = TR * 64;
  static constexpr int ELEMS = 2 * PLANE;
  static __device__ __forceinline__ int offset(int row, int d) { return ptkv::plane_offset(row, d) >> 1; }
  static __device__ __forceinline__ void store4(unsigned short *__restrict__ s, int row, int d, const float4 &v, float sc) {
    uint2 t1, t2;
    split_quad_f16(v.x, v.y, v.z, v.w, sc, sc, sc, sc, t1, t2);
    const int off = offset(row, d);
    *reinterpret_cast<uint2 *>(s + off) = t1;
    *reinterpret_cast<uint2 *>(s + PLANE + off) = t2;
  }
  static __device__ __forceinline__ void frag_rows(const unsigned short *__restrict__ s, int step, int lane, f16x8 (&f)[2]) {
    const unsigned short *q = s + offset(lane & 31, 16 * step + 8 * (lane >> 5));
#pragma unroll
    for (int t = 0; t < 2; ++t) f[t] = *reinterpret_cast<const f16x8 *>(q + t * PLANE);
  }
  static __device__ __forceinline__ void frag_cols(const unsigned short *__restrict__ s, int kb, int d0, int lane,
                                                   f16x8 (&f)[2]) {
    const int q16 = lane & 15;
    const int row = kb + 4 * (lane >> 5) + (q16 >> 2);
    const unsigned short *q0 = s + offset(row, d0 + (lane & 16) + 4 * (q16 & 3));
    const unsigned short *q1 = s + offset(row + 8, d0 + (lane & 16) + 4 * (q16 & 3));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(q0 + t * PLANE));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(q1 + t * PLANE));
      const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      f[t] = __builtin_bit_cast(f16x8, both);
    }
  }
};

// ---- a K / V tile that arrives PRE-SPLIT (kv_format.h): [plane][32 rows][64 f16], chunk-swizzled, written into LDS by LDS-DMA
typedef __attribute__((address_space(1))) const void *kv_gptr_t;
typedef __attribute__((address_space(3))) void *kv_lptr_t;
struct KvTile {
  // 64 lanes x 16 B of global memory (g already holds the lane's + 16 lane) -> 1 KiB of LDS at l (wavefront-uniform)
  static __device__ __forceinline__ void dma16(const char *g, char *l) {
    __builtin_amdgcn_global_load_lds((kv_gptr_t)g, (kv_lptr_t)l, 16, 0, 0);
  }
  // lane l holds tile row (l & 31), d = 16 step + 8 (l >> 5) + 0..7   (Tile2::frag_rows)
  static __device__ __forceinline__ void frag_rows(const char *__restrict__ s, int step, int lane, f16x8 (&f)[2]) {
    const char *q = s + ptkv::plane_offset(lane & 31, 16 * step + 8 * (lane >> 5));
#pragma unroll
    for (int t = 0; t < 2; ++t) f[t] = *reinterpret_cast<const f16x8 *>(q + t * ptkv::PLANE_BYTES);
  }
  // the transposed fragment of Tile2::frag_cols: column d0 + (l & 31) of tile rows kb + 4 (l >> 5) + {0..3} and + 8
  static __device__ __forceinline__ void frag_cols(const char *__restrict__ s, int kb, int d0, int lane, f16x8 (&f)[2]) {
    const int q16 = lane & 15;
    const int row = kb + 4 * (lane >> 5) + (q16 >> 2), d = d0 + (lane & 16) + 4 * (q16 & 3);
    const char *q0 = s + ptkv::plane_offset(row, d), *q1 = s + ptkv::plane_offset(row + 8, d);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(q0 + t * ptkv::PLANE_BYTES));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(q1 + t * ptkv::PLANE_BYTES));
      const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      f[t] = __builtin_bit_cast(f16x8, both);
    }
  }
};
// one row of the planes as the lane's B operand (load_row_scaled for pre-split K / V): lane (l31, lh) holds d = 16 s + 8 lh + 0..7
// of token `tok`; returns the INVERSE scale of the row's group of four tokens
template <int KS>
__device__ __forceinline__ float load_row_planes(const char *__restrict__ planes, const float *__restrict__ inv, size_t hc_tiles,
                                                 int tok, bool ok, int lh, f16x8 (&f)[KS][2]) {
  const size_t tile = hc_tiles + (size_t)(tok >> 5);
  const int r = tok & 31;
  const char *base = planes + tile * ptkv::TILE_BYTES;
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const uint4 v = ok ? *reinterpret_cast<const uint4 *>(base + t * ptkv::PLANE_BYTES + ptkv::plane_offset(r, 16 * s + 8 * lh))
                         : make_uint4(0u, 0u, 0u, 0u);
      f[s][t] = __builtin_bit_cast(f16x8, v);
    }
  return ok ? inv[tile * 8 + ptkv::group_slot(r >> 2)] : 1.f;
}

// 32 rows x DK floats of a [*, ld] matrix: global -> registers (one float4 per thread) -> scaled f16 planes in LDS.
// A group of four rows is held by 4 DK / 4 adjacent lanes (one wavefront for DK = 64, half of one for DK = 32); its
// inverse scale goes to inv[(g & 1) * 4 + (g >> 1)], g = row >> 2, so that a lane half reads ITS four groups
// (g = 2 j + lh) as one float4.  Loads are unconditional (row clamped); rows beyond nrows are zeroed when stored.
template <int DK, int NW>
struct StageGeo {
  static constexpr int CPR = DK / 4;               // float4 per tile row
  static constexpr int RPP = 64 * NW / CPR;        // tile rows the workgroup covers with one float4 per thread
  static constexpr int NI = RPP >= TR ? 1 : TR / RPP;  // float4 per thread and tile
};
template <int NI>
struct TileOff {
  uint32_t o[NI];
};
template <int DK, int NW, typename TILE = Tile2>
struct Stage {
  using G = StageGeo<DK, NW>;
  static constexpr int CPR = G::CPR, NI = G::NI;
  float4 v[NI];
  __device__ __forceinline__ void load(const float *__restrict__ base, const TileOff<NI> &off) {
#pragma unroll
    for (int i = 0; i < NI; ++i) v[i] = *reinterpret_cast<const float4 *>(base + off.o[i]);
  }
  __device__ __forceinline__ void store(unsigned short *__restrict__ s, float *__restrict__ inv, int row0, int nrows,
                                        int tid) const {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int row = tid / CPR + i * G::RPP;
      const bool ok = row < TR && row0 + row < nrows;
      const float4 x = make_float4(ok ? v[i].x : 0.f, ok ? v[i].y : 0.f, ok ? v[i].z : 0.f, ok ? v[i].w : 0.f);
      const uint32_t amax = group_umax<4 * CPR>(umax4(x));
      const uint32_t sbits = pt_row_scale_bits(amax);
      if (row < TR) {  // (wavefront-uniform)
        TILE::store4(s, row, (tid % CPR) * 4, x, __uint_as_float(sbits));
        if ((tid % (4 * CPR)) == 0) {
          const int g = row >> 2;
          inv[(g & 1) * 4 + (g >> 1)] = __uint_as_float((254u << 23) - sbits);
        }
      }
    }
  }
};

// Element offsets of a thread's float4 in consecutive 32-row tiles of a [nrows, ld] block (rows clamped to the last one):
// an add and a min per load instead of a 64-bit multiply; tiles that share rows (K and V, same ld) share them.
template <int DK, int NW>
struct TileRows {
  using G = StageGeo<DK, NW>;
  uint32_t u, lim, col, step, pass;
  __device__ __forceinline__ void init(int ld, int nrows, int tid, int first_row = 0) {
    u = (uint32_t)(first_row + min(tid / G::CPR, TR - 1)) * (uint32_t)ld;
    lim = (uint32_t)(nrows - 1) * (uint32_t)ld;
    col = (uint32_t)(tid % G::CPR) * 4u;
    step = (uint32_t)TR * (uint32_t)ld;
    pass = (uint32_t)G::RPP * (uint32_t)ld;
  }
  __device__ __forceinline__ TileOff<G::NI> next() {
    TileOff<G::NI> off;
#pragma unroll
    for (int i = 0; i < G::NI; ++i) off.o[i] = min(u + (uint32_t)i * pass, lim) + col;
    u += step;
    return off;
  }
};

// one row of a [*, ld] matrix as the lane's B operand: lane (l31, lh) holds d = 16 s + 8 lh + 0..7; the row is scaled by
// the power of two of its own maximum (returned: the INVERSE scale)
template <int KS>
__device__ __forceinline__ float load_row_scaled(const float *__restrict__ base, int ld, int row, bool ok, int lh,
                                                 f16x8 (&f)[KS][2]) {
  float x[KS][8];
  uint32_t amax = 0;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const float *p = base + (size_t)row * ld + 16 * s + 8 * lh;
    const float4 a = ok ? *reinterpret_cast<const float4 *>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b = ok ? *reinterpret_cast<const float4 *>(p + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    x[s][0] = a.x; x[s][1] = a.y; x[s][2] = a.z; x[s][3] = a.w;
    x[s][4] = b.x; x[s][5] = b.y; x[s][6] = b.z; x[s][7] = b.w;
    amax = max(amax, max(umax4(a), umax4(b)));
  }
  amax = max(amax, (uint32_t)__shfl_xor((int)amax, 32, 64));
  const uint32_t sbits = pt_row_scale_bits(amax);
  const float sc = __uint_as_float(sbits);
#pragma unroll
  for (int s = 0; s < KS; ++s) split8g(x[s], sc, sc, f[s]);
  return __uint_as_float((254u << 23) - sbits);
}

// On-line common power of two of a register operand (dS) whose largest magnitude of this tile is m (equal in both lane
// halves): returns the factor the accumulators have to be multiplied by (1 when the scale stands) and updates `bscale`
// so that m bscale < 2^14 - the products of the tile then fit f16 with a bit to spare.
__device__ __forceinline__ float online_scale(float m, float &bscale) {
  float ratio = 1.f;
  if (m * bscale >= TWO14) {  // also taken when the product overflows
    const uint32_t e = __float_as_uint(m) >> 23;                // >= 21 here, <= 254 for finite m
    const float nb = __uint_as_float((264u - min(e, 254u)) << 23);  // m nb in [2^10, 2^11)
    ratio = nb * inv_pow2(bscale);
    bscale = nb;
  }
  return ratio;
}
// Row scales of dqkv for the f16x2 products that consume it (ptamd_gemm a_scale / uniform scale), so that no pass over
// dqkv is needed: a lane pair holds one row of one head's block; the f16x2 row scale is a decreasing function of the
// row maximum, so the blocks of a row combine by atomicMin on targets the caller preset to 0x7F000000.  row_min[0..3]
// receive the smallest scale of all rows (the uniform scale of dqkv as an operand of the weight-gradient product).
__device__ __forceinline__ void publish_row_scale(float amax, bool ok, int row, int tid, uint32_t *__restrict__ row_scale,
                                                  uint32_t *__restrict__ row_min, unsigned int *__restrict__ s_min) {
  const int lane = tid & 63;
  amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
  uint32_t sb = ok ? pt_row_scale_bits(__float_as_uint(amax)) : 0x7F000000u;
  if (ok && lane < 32) atomicMin(row_scale + row, sb);
  if (row_min) {  // one global atomic per workgroup and copy: the wavefronts meet in LDS first
    if (tid == 0) *s_min = 0x7F000000u;
    __syncthreads();
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sb = min(sb, (uint32_t)__shfl_xor((int)sb, o, 64));
    if (lane == 0) atomicMin(s_min, sb);
    __syncthreads();
    if (tid < 4) atomicMin(row_min + tid, *s_min);
  }
}

constexpr float BSCALE0 = 1.329227995784916e36f;  // 2^120

constexpr int BUF = 2 * Tile2::ELEMS;  // f16 elements of one {A, B} tile buffer
constexpr size_t ATTN_LDS = (size_t)2 * BUF * sizeof(unsigned short);

// =================================================================================================== forward
#ifndef PT_ATTN_FWD_WAVES
#define PT_ATTN_FWD_WAVES 2   // wavefronts per SIMD the forward kernel is compiled for (4 = 128 VGPRs: 27 spilled, measured in round 4)
#endif
// TILE: the LDS image of a staged tile - Tile2 (padded rows) or Tile2U (unpadded, swizzled: what lets PARTS = 4 fit)
template <int DK, int NW, int PARTS, typename TILE = Tile2>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(PT_ATTN_FWD_WAVES, PT_ATTN_FWD_WAVES))) void attn_fwd_f16x2_kernel(
    const float *__restrict__ qkv, const int64_t *__restrict__ seq, int L, int H, float p_drop, uint64_t seed,
    uint32_t stream_id, float *__restrict__ out, float *__restrict__ lse, uint32_t *__restrict__ keep_bits) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  __shared__ __attribute__((aligned(16))) float sBias[2][PARTS][TR];
  __shared__ __attribute__((aligned(16))) float sInvK[2][PARTS][8], sInvV[2][PARTS][8];
  constexpr int NG = NW / PARTS;  // query groups of the workgroup
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int grp = PARTS == 1 ? wave : wave % NG, part = PARTS == 1 ? 0 : wave / NG;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * (32 * NG) + grp * 32;
  const int D = H * DK, D3 = 3 * D;
  const float *base = qkv + (size_t)b * L * D3 + h * DK;  // Q block of this head; K at +D, V at +2D
  const int64_t *sq = seq + (size_t)b * L;
  const int q = q0 + l31;
  const bool q_ok = q < L;
  constexpr int KS = DK / 16, NT = DK / 32;
  const float scale = DK == 64 ? 0.125f : 0.17677669529663687f;  // 1 / sqrt(dk)
  const AttnDrop dk_ = make_attn_drop(seed, stream_id, (uint32_t)(b * H + h), p_drop);
  const uint32_t q_part = attn_q_part(dk_, (uint32_t)q);

  f16x8 qf[KS][2];
  const float iq = load_row_scaled<KS>(base, D3, min(q, L - 1), q_ok, lh, qf);
  const float cq = scale * LOG2E * iq;  // scores leave the accumulators in log2 units: exp(x) = exp2(x log2 e)

  f32x16 o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;  // running maximum in log2 units
  float v_run = 0.f;                     // largest inverse V group scale so far (a power of two; wavefront-uniform)

  constexpr int TBUF = 2 * TILE::ELEMS;  // f16 elements of one {K, V} tile buffer
  Stage<DK, NW, TILE> stK[PARTS], stV[PARTS];
  const int ntiles = ((L + TR - 1) / TR + PARTS - 1) / PARTS;  // tiles of one part: part p walks tiles p ntiles ..
  auto tile = [&](int buf, int pt) __attribute__((always_inline)) { return smem + (buf * PARTS + pt) * TBUF; };
  // key mask of a tile as an ADDITIVE term of the soft-max argument (0 or -inf per key), read back one float4 per register
  // quadruple: no bit extraction and no select per element
  auto publish_mask = [&](int k0, int buf, int pt) __attribute__((always_inline)) {
    if (tid < TR) {
      const int key = k0 + tid;
      sBias[buf][pt][tid] = (key < L && sq[key < L ? key : 0] != PTAMD_PAD_ID) ? 0.f : -INFINITY;
    }
  };
  TileRows<DK, NW> rows[PARTS];
  Stage<DK, NW, TILE> nxK[PARTS], nxV[PARTS];  // loads run two tiles ahead of the arithmetic (attention_split.hip)
#pragma unroll
  for (int pt = 0; pt < PARTS; ++pt) {
    rows[pt].init(D3, L, tid, pt * ntiles * TR);
    const auto toff = rows[pt].next();
    stK[pt].load(base + D, toff);
    stV[pt].load(base + 2 * D, toff);
    stK[pt].store(tile(0, pt), sInvK[0][pt], pt * ntiles * TR, L, tid);
    stV[pt].store(tile(0, pt) + TILE::ELEMS, sInvV[0][pt], pt * ntiles * TR, L, tid);
    publish_mask(pt * ntiles * TR, 0, pt);
    const auto toff2 = rows[pt].next();
    stK[pt].load(base + D, toff2);
    stV[pt].load(base + 2 * D, toff2);
  }
  __syncthreads();

  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = (part * ntiles + kt) * TR, cur = kt & 1;
    const bool more = kt + 1 < ntiles;
    const unsigned short *sK = tile(cur, part), *sV = sK + TILE::ELEMS;
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt) {
      const auto toff = rows[pt].next();
      nxK[pt].load(base + D, toff);
      nxV[pt].load(base + 2 * D, toff);
    }
    const float4 ik4 = *reinterpret_cast<const float4 *>(&sInvK[cur][part][4 * lh]);  // the four key groups of this lane half
    const float4 iva = *reinterpret_cast<const float4 *>(&sInvV[cur][part][0]), ivb = *reinterpret_cast<const float4 *>(&sInvV[cur][part][4]);
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int st = 0; st < KS; ++st) {  // S^T[key][q] = K Q^T (scaled operands)
      f16x8 kf[2];
      TILE::frag_rows(sK, st, lane, kf);
      s = mfma3(kf, qf[st], s);
    }
    if (more) {  // scale + split + store the next tile(s) into the other buffer while the soft-max runs
#pragma unroll
      for (int pt = 0; pt < PARTS; ++pt) {
        const int kn = (pt * ntiles + kt + 1) * TR;
        stK[pt].store(tile(cur ^ 1, pt), sInvK[cur ^ 1][pt], kn, L, tid);
        stV[pt].store(tile(cur ^ 1, pt) + TILE::ELEMS, sInvV[cur ^ 1][pt], kn, L, tid);
        publish_mask(kn, cur ^ 1, pt);
      }
    }
    const float cu[4] = {cq * ik4.x, cq * ik4.y, cq * ik4.z, cq * ik4.w};
    float mt = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 b4 = *reinterpret_cast<const float4 *>(&sBias[cur][part][8 * j + 4 * lh]);
      const float bias[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {  // (s cu is finite - a masked group of zero rows has s = cu = 0 - so the sum is -inf, not NaN)
        s[4 * j + e] = fmaf(s[4 * j + e], cu[j], bias[e]);
        mt = fmaxf(mt, s[4 * j + e]);
      }
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(s[r] - m_safe);
      ps += s[r];
    }
    l_run = l_run * alpha + ps;
    m_run = m_new;
    // common scale of the V groups: O accumulates  sum P (iv_g / v_run) 2^14 * (V sv_g)  =  (2^14 / v_run) sum P V
    const float vt = fmaxf(fmaxf(fmaxf(iva.x, iva.y), fmaxf(iva.z, iva.w)), fmaxf(fmaxf(ivb.x, ivb.y), fmaxf(ivb.z, ivb.w)));
    float resc = alpha;
    if (vt > v_run) {
      resc *= v_run * inv_pow2(vt);
      v_run = vt;
    }
    if (__builtin_amdgcn_ballot_w64(resc != 1.f)) {  // wave-uniform: after the first tiles neither maximum moves often
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {  // scalar multiplies, kept apart: packed f32 VALU stalls the matrix pipe
          float v = o[t][r] * resc;
          asm volatile("" : "+v"(v));
          o[t][r] = v;
        }
    }
    const float vn = inv_pow2(v_run);
    const float4 ivh = lh ? ivb : iva;
    const float fv[4] = {ivh.x * TWO14 * vn, ivh.y * TWO14 * vn, ivh.z * TWO14 * vn, ivh.w * TWO14 * vn};  // <= 2^14
    if (p_drop > 0.f) {  // the 1 / (1 - p) is applied to O at the end
      if (keep_bits) {   // (uniform) the decisions also go out for the fused backward kernel: attn_dropout.h
        const uint32_t word = attn_drop_keys_in_rows_export(dk_, q_part, k0, lh, s);
        const int lk = (L + 31) & ~31;
        if (lane < 32 && k0 + lane < lk && q0 < lk)   // (a workgroup's last wavefronts may hold no query at all)
          keep_bits[((size_t)(b * H + h) * (lk >> 5) + (q0 >> 5)) * lk + k0 + lane] = word;
      } else {
        attn_drop_keys_in_rows(dk_, q_part, k0, lh, s);
      }
    }
    // O^T[d][q] += V^T[d][key] P^T[key][q]: the accumulator rows of s are already in the k order of frag_cols
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float x[8] = {s[8 * m], s[8 * m + 1], s[8 * m + 2], s[8 * m + 3], s[8 * m + 4], s[8 * m + 5], s[8 * m + 6], s[8 * m + 7]};
      f16x8 pf[2];
      split8g(x, fv[2 * m], fv[2 * m + 1], pf);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f16x8 vf[2];
        TILE::frag_cols(sV, 16 * m, 32 * t, lane, vf);
        o[t] = mfma3(vf, pf, o[t]);
      }
    }
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt) {
      stK[pt] = nxK[pt];
      stV[pt] = nxV[pt];
    }
    __syncthreads();  // the other buffer is complete; nobody reads this one any more
  }

  if (PARTS > 1) {
    // the later key ranges hand their soft-max state (running maximum, row sum of the lane half, V scale) and their accumulators
    // to the wavefront of the same queries that walked the first range: [part - 1][group][slot][lane] floats in the tile area;
    // it takes them in part order (fixed: deterministic)
    constexpr int XCH = (NT * 16 + 3) * 64;
    float *xch = reinterpret_cast<float *>(smem) + ((part > 0 ? part - 1 : 0) * NG + grp) * XCH + lane;
    if (part > 0) {
      xch[0] = m_run;
      xch[64] = l_run;
      xch[128] = v_run;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(3 + t * 16 + r) * 64] = o[t][r];
    }
    __syncthreads();
    if (part > 0) return;
#pragma unroll
    for (int k = 0; k < PARTS - 1; ++k) {
      const float *x = xch + (size_t)k * NG * XCH;
      const float m1 = x[0], l1 = x[64], v1 = x[128];
      const float m = fmaxf(m_run, m1), ms = m == -INFINITY ? 0.f : m;
      const float a0 = __builtin_amdgcn_exp2f(m_run - ms), a1 = __builtin_amdgcn_exp2f(m1 - ms);
      const float vm = fmaxf(v_run, v1), ivm = inv_pow2(vm);  // common V scale (powers of two; 0 only if both are)
      const float f0 = a0 * (v_run * ivm), f1 = a1 * (v1 * ivm);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = o[t][r] * f0 + x[(3 + t * 16 + r) * 64] * f1;
      l_run = l_run * a0 + l1 * a1;
      m_run = m;
      v_run = vm;
    }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  // soft-max normalisation, dropout scale and the common V scale in one factor
  const float inv = (p_drop > 0.f ? dk_.ks : 1.f) * v_run * INV_TWO14 / l_tot;
  if (q_ok) {
    float *op = out + (size_t)(b * L + q) * D + h * DK;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = t * 32 + 8 * g + 4 * lh;
        *reinterpret_cast<float4 *>(op + d) =
            make_float4(o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
      }
    if (lh == 0) lse[((size_t)b * H + h) * L + q] = (m_run + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The forward kernel on PRE-SPLIT K / V (kv_format.h; dk = 64, 8 wavefronts = 256 queries per workgroup): what the QKV
// product's epilogue wrote is byte for byte the LDS image of a tile, so a stage of K and V is filled by 1-KiB LDS-DMA pieces
// (two per wavefront and 32-key tile) instead of two float4 loads, a group maximum, a split and two ds_write per thread and
// tile - the staging that was a fifth of the kernel's VALU instructions (profiles/tools/isa_mix.py: 570 -> 491 per tile and
// wavefront, all paths).  Same scaling groups, same split arithmetic, same soft-max, same dropout decisions: bit-identical to
// attn_fwd_f16x2_kernel<64, 8, 1> on the fp32 K / V those planes were made from.
// A stage is TPS tiles (TPS x 32 keys): THREE stage buffers, the pieces of stage s + 2 issued at the end of stage s and a
// counted wait (this wavefront's pieces of that stage may stay in flight) in front of a raw s_barrier - HBM latency is two
// stages of arithmetic, and there is ONE barrier per stage: with TPS = 2 half as many as the fp32 kernel, whose eight
// wavefronts meet at every tile and so keep walking its phases (matrix products, soft-max, matrix products) in lock step.
// Q stays an fp32 row per lane (one load per workgroup).
template <int TPS>
struct KvpGeo {
  static constexpr int STAGE = TPS * 2 * ptkv::TILE_BYTES;   // per tile: K tile, V tile
  static constexpr size_t LDS = 3 * (size_t)STAGE;
};
template <int TPS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(PT_ATTN_FWD_WAVES, PT_ATTN_FWD_WAVES))) void attn_fwd_kvp_f16x2_kernel(
    const float *__restrict__ qkv, const char *__restrict__ kvp, const float *__restrict__ kv_inv, int kv_nt,
    const int64_t *__restrict__ seq, int L, int H, float p_drop, uint64_t seed, uint32_t stream_id, float *__restrict__ out,
    float *__restrict__ lse, uint32_t *__restrict__ keep_bits) {
  constexpr int DK = 64, KS = DK / 16, NT = DK / 32, STAGE = KvpGeo<TPS>::STAGE;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  __shared__ __attribute__((aligned(16))) float sBias[2][TPS][TR];
  __shared__ __attribute__((aligned(16))) float sInvK[2][TPS][8], sInvV[2][TPS][8];
  char *const stage0 = reinterpret_cast<char *>(smem);
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 256 + wave * 32;
  const int D = H * DK, D3 = 3 * D;
  const float *base = qkv + (size_t)b * L * D3 + h * DK;  // Q block of this head
  const int64_t *sq = seq + (size_t)b * L;
  const int q = q0 + l31;
  const bool q_ok = q < L;
  const float scale = 0.125f;  // 1 / sqrt(dk)
  const AttnDrop dk_ = make_attn_drop(seed, stream_id, (uint32_t)(b * H + h), p_drop);
  const uint32_t q_part = attn_q_part(dk_, (uint32_t)q);
  // tiles of this (protein, head): global tile gt0 + kt of K (which = 0) and V (which = 1); L is a multiple of 32 here
  const int ntiles = L >> 5, nstages = (ntiles + TPS - 1) / TPS, gt0 = (b * L) >> 5;
  const size_t tk = ptkv::tile_index(0, h, gt0, H, kv_nt), tv = ptkv::tile_index(1, h, gt0, H, kv_nt);
  const char *const gk = kvp + tk * ptkv::TILE_BYTES + wave * 1024 + lane * 16;   // this wavefront's piece of a K / V tile
  const char *const gv = kvp + tv * ptkv::TILE_BYTES + wave * 1024 + lane * 16;
  // the pieces of stage st (2 per tile: K, V) into stage buffer buf; returns nothing - the number issued is 2 x tiles of the stage
  auto issue = [&](int st, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < TPS; ++u) {
      const int kt = st * TPS + u;
      if (kt < ntiles) {   // (uniform)
        KvTile::dma16(gk + (size_t)kt * ptkv::TILE_BYTES, stage0 + buf * STAGE + u * 2 * ptkv::TILE_BYTES + wave * 1024);
        KvTile::dma16(gv + (size_t)kt * ptkv::TILE_BYTES, stage0 + buf * STAGE + (u * 2 + 1) * ptkv::TILE_BYTES + wave * 1024);
      }
    }
  };
  // inverse group scales (8 + 8 floats) and the key mask of the tiles of a stage: loaded at the top of the stage before, published
  // at its end.  (The loaded words are not touched before publish(): the wait for them then sits at the end of the stage, where
  // the pieces of the next stage have to be in anyway - used at once, wavefront 0 would wait vmcnt(0) at the top of every stage.)
  float r_inv = 0.f;
  int64_t r_seq = 0;
  auto fetch_small = [&](int st) __attribute__((always_inline)) {
    const int u = tid >> 5, kt = st * TPS + u;               // thread tid serves tile u = tid / 32 of the stage
    if (tid < 32 * TPS && kt < ntiles) {
      r_seq = sq[kt * TR + (tid & 31)];
      if ((tid & 31) < 16) r_inv = kv_inv[(((tid & 31) < 8 ? tk : tv) + kt) * 8 + (tid & 7)];
    }
  };
  auto publish = [&](int slot) __attribute__((always_inline)) {
    const int u = tid >> 5, t31 = tid & 31;
    if (tid < 32 * TPS) {
      if (t31 < 8) sInvK[slot][u][t31] = r_inv;
      else if (t31 < 16) sInvV[slot][u][t31 - 8] = r_inv;
      sBias[slot][u][t31] = r_seq != PTAMD_PAD_ID ? 0.f : -INFINITY;
    }
  };

  issue(0, 0);
  fetch_small(0);
  f16x8 qf[KS][2];
  const float iq = load_row_scaled<KS>(base, D3, min(q, L - 1), q_ok, lh, qf);
  const float cq = scale * LOG2E * iq;
  f32x16 o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f, v_run = 0.f;
  publish(0);
  if (nstages > 1) {
    issue(1, 1);
    // stage 0 is in; the pieces of stage 1 (2 per tile it holds) may stay in flight
    if (TPS == 1 || ntiles >= 2 * TPS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * TPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int buf = 0;
  for (int st = 0; st < nstages; ++st) {
    const int cur = st & 1;
    const bool more = st + 1 < nstages, more2 = st + 2 < nstages;
    const int nbuf2 = buf >= 1 ? buf - 1 : 2;                // (buf + 2) % 3: read last in stage st - 1
    if (more) fetch_small(st + 1);
#pragma unroll
    for (int u = 0; u < TPS; ++u) {
      const int kt = st * TPS + u, k0 = kt * TR;
      if (kt >= ntiles) break;   // (uniform; only the last stage can be short)
      const char *sK = stage0 + buf * STAGE + u * 2 * ptkv::TILE_BYTES, *sV = sK + ptkv::TILE_BYTES;
      const float4 ik4 = *reinterpret_cast<const float4 *>(&sInvK[cur][u][4 * lh]);
      const float4 iva = *reinterpret_cast<const float4 *>(&sInvV[cur][u][0]), ivb = *reinterpret_cast<const float4 *>(&sInvV[cur][u][4]);
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int k = 0; k < KS; ++k) {  // S^T[key][q] = K Q^T (scaled operands)
        f16x8 kf[2];
        KvTile::frag_rows(sK, k, lane, kf);
        s = mfma3(kf, qf[k], s);
      }
      const float cu[4] = {cq * ik4.x, cq * ik4.y, cq * ik4.z, cq * ik4.w};
      float mt = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 b4 = *reinterpret_cast<const float4 *>(&sBias[cur][u][8 * j + 4 * lh]);
        const float bias[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[4 * j + e] = fmaf(s[4 * j + e], cu[j], bias[e]);
          mt = fmaxf(mt, s[4 * j + e]);
        }
      }
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float m_new = fmaxf(m_run, mt);
      const float m_safe = m_new == -INFINITY ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(s[r] - m_safe);
        ps += s[r];
      }
      l_run = l_run * alpha + ps;
      m_run = m_new;
      const float vt = fmaxf(fmaxf(fmaxf(iva.x, iva.y), fmaxf(iva.z, iva.w)), fmaxf(fmaxf(ivb.x, ivb.y), fmaxf(ivb.z, ivb.w)));
      float resc = alpha;
      if (vt > v_run) {
        resc *= v_run * inv_pow2(vt);
        v_run = vt;
      }
      if (__builtin_amdgcn_ballot_w64(resc != 1.f)) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = o[t][r] * resc;
            asm volatile("" : "+v"(v));
            o[t][r] = v;
          }
      }
      const float vn = inv_pow2(v_run);
      const float4 ivh = lh ? ivb : iva;
      const float fv[4] = {ivh.x * TWO14 * vn, ivh.y * TWO14 * vn, ivh.z * TWO14 * vn, ivh.w * TWO14 * vn};
      if (p_drop > 0.f) {
        if (keep_bits) {
          const uint32_t word = attn_drop_keys_in_rows_export(dk_, q_part, k0, lh, s);
          if (lane < 32 && q0 < L)
            keep_bits[((size_t)(b * H + h) * (L >> 5) + (q0 >> 5)) * L + k0 + lane] = word;
        } else {
          attn_drop_keys_in_rows(dk_, q_part, k0, lh, s);
        }
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const float x[8] = {s[8 * m], s[8 * m + 1], s[8 * m + 2], s[8 * m + 3], s[8 * m + 4], s[8 * m + 5], s[8 * m + 6], s[8 * m + 7]};
        f16x8 pf[2];
        split8g(x, fv[2 * m], fv[2 * m + 1], pf);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          f16x8 vf[2];
          KvTile::frag_cols(sV, 16 * m, 32 * t, lane, vf);
          o[t] = mfma3(vf, pf, o[t]);
        }
      }
    }
    // The pieces of stage st + 2 are this wavefront's LAST memory operations of the stage (behind the decision stores, the small
    // loads and every LDS read): the counted wait then leaves exactly them in flight - stage st + 1, issued a stage ago, has
    // landed; behind the barrier everybody is done with this buffer.
    if (more) publish(cur ^ 1);
    if (more2) issue(st + 2, nbuf2);
    if (more2 && (TPS == 1 || (st + 3) * TPS <= ntiles)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * TPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    buf = buf == 2 ? 0 : buf + 1;
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = (p_drop > 0.f ? dk_.ks : 1.f) * v_run * INV_TWO14 / l_tot;
  if (q_ok) {
    float *op = out + (size_t)(b * L + q) * D + h * DK;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = t * 32 + 8 * g + 4 * lh;
        *reinterpret_cast<float4 *>(op + d) =
            make_float4(o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
      }
    if (lh == 0) lse[((size_t)b * H + h) * L + q] = (m_run + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
  }
}

// =================================================================================================== backward
// dQ: same decomposition as the forward kernel.  Also computes delta[q] = sum_d dO[q,d] O[q,d] and publishes it for the
// dK/dV kernel, which runs after this one on the same stream.
template <int DK, int NW, int PARTS>
__global__ __launch_bounds__(64 * NW, 1) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dq_f16x2_kernel(
    const float *__restrict__ qkv, const int64_t *__restrict__ seq, const float *__restrict__ o_fwd,
    const float *__restrict__ d_o, const float *__restrict__ lse, float *__restrict__ delta, int L, int H, float p_drop,
    uint64_t seed, uint32_t stream_id, float *__restrict__ dqkv, uint32_t *__restrict__ row_scale,
    uint32_t *__restrict__ row_min) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  __shared__ __attribute__((aligned(16))) float sBias[2][PARTS][TR];
  __shared__ unsigned int sMin;
  __shared__ __attribute__((aligned(16))) float sInvK[2][PARTS][8], sInvV[2][PARTS][8];
  constexpr int NG = NW / PARTS;  // query groups of the workgroup
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int grp = PARTS == 1 ? wave : wave % NG, part = PARTS == 1 ? 0 : wave / NG;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * (32 * NG) + grp * 32;
  const int D = H * DK, D3 = 3 * D;
  const float *base = qkv + (size_t)b * L * D3 + h * DK;
  const int64_t *sq = seq + (size_t)b * L;
  const int q = q0 + l31, qc = min(q, L - 1);
  const bool q_ok = q < L;
  constexpr int KS = DK / 16, NT = DK / 32;
  const float scale = DK == 64 ? 0.125f : 0.17677669529663687f;
  const AttnDrop dk_ = make_attn_drop(seed, stream_id, (uint32_t)(b * H + h), p_drop);
  const uint32_t q_part = attn_q_part(dk_, (uint32_t)q);

  f16x8 qf[KS][2], gf[KS][2];
  const float iq = load_row_scaled<KS>(base, D3, qc, q_ok, lh, qf);
  const float ig = load_row_scaled<KS>(d_o + (size_t)b * L * D + h * DK, D, qc, q_ok, lh, gf);
  const float my_lse2 = q_ok ? lse[((size_t)b * H + h) * L + q] * LOG2E : 0.f;
  float my_delta = 0.f;
  {  // each lane half holds half of the d of its query's row
    const float *gp = d_o + ((size_t)b * L + qc) * D + h * DK, *op = o_fwd + ((size_t)b * L + qc) * D + h * DK;
#pragma unroll
    for (int st = 0; st < KS; ++st)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float4 g4 = *reinterpret_cast<const float4 *>(gp + 16 * st + 8 * lh + 4 * j);
        const float4 o4 = *reinterpret_cast<const float4 *>(op + 16 * st + 8 * lh + 4 * j);
        my_delta += g4.x * o4.x + g4.y * o4.y + g4.z * o4.z + g4.w * o4.w;
      }
    my_delta += __shfl_xor(my_delta, 32, 64);
    if (!q_ok) my_delta = 0.f;
    if (q_ok && lh == 0 && part == 0) delta[((size_t)b * H + h) * L + q] = my_delta;
  }
  const float cq = scale * LOG2E * iq;
  const float gq = ig * (p_drop > 0.f ? dk_.ks : 1.f);

  f32x16 dq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;
  float bscale = BSCALE0;  // common power of two of the dS operand (per query = per accumulator column)

  Stage<DK, NW> stK[PARTS], stV[PARTS];
  const int ntiles = ((L + TR - 1) / TR + PARTS - 1) / PARTS;  // tiles of one part: part p walks tiles p ntiles ..
  auto tile = [&](int buf, int pt) __attribute__((always_inline)) { return smem + (buf * PARTS + pt) * BUF; };
  // key mask of a tile as an ADDITIVE term of the soft-max argument (0 or -inf per key), read back one float4 per register
  // quadruple: no bit extraction and no select per element
  auto publish_mask = [&](int k0, int buf, int pt) __attribute__((always_inline)) {
    if (tid < TR) {
      const int key = k0 + tid;
      sBias[buf][pt][tid] = (key < L && sq[key < L ? key : 0] != PTAMD_PAD_ID) ? 0.f : -INFINITY;
    }
  };
  TileRows<DK, NW> rows[PARTS];
  Stage<DK, NW> nxK[PARTS], nxV[PARTS];
#pragma unroll
  for (int pt = 0; pt < PARTS; ++pt) {
    rows[pt].init(D3, L, tid, pt * ntiles * TR);
    const auto toff = rows[pt].next();
    stK[pt].load(base + D, toff);
    stV[pt].load(base + 2 * D, toff);
    stK[pt].store(tile(0, pt), sInvK[0][pt], pt * ntiles * TR, L, tid);
    stV[pt].store(tile(0, pt) + Tile2::ELEMS, sInvV[0][pt], pt * ntiles * TR, L, tid);
    publish_mask(pt * ntiles * TR, 0, pt);
    const auto toff2 = rows[pt].next();
    stK[pt].load(base + D, toff2);
    stV[pt].load(base + 2 * D, toff2);
  }
  __syncthreads();

  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = (part * ntiles + kt) * TR, cur = kt & 1;
    const bool more = kt + 1 < ntiles;
    const unsigned short *sK = tile(cur, part), *sV = sK + Tile2::ELEMS;
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt) {
      const auto toff = rows[pt].next();
      nxK[pt].load(base + D, toff);
      nxV[pt].load(base + 2 * D, toff);
    }
    const float4 ik4 = *reinterpret_cast<const float4 *>(&sInvK[cur][part][4 * lh]);
    const float4 iv4 = *reinterpret_cast<const float4 *>(&sInvV[cur][part][4 * lh]);
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int st = 0; st < KS; ++st) {
      f16x8 kf[2], vf[2];
      Tile2::frag_rows(sK, st, lane, kf);
      Tile2::frag_rows(sV, st, lane, vf);
      s = mfma3(kf, qf[st], s);     // S^T[key][q]
      dp = mfma3(vf, gf[st], dp);   // dP^T[key][q] = V dO^T
    }
    if (more) {
#pragma unroll
      for (int pt = 0; pt < PARTS; ++pt) {
        const int kn = (pt * ntiles + kt + 1) * TR;
        stK[pt].store(tile(cur ^ 1, pt), sInvK[cur ^ 1][pt], kn, L, tid);
        stV[pt].store(tile(cur ^ 1, pt) + Tile2::ELEMS, sInvV[cur ^ 1][pt], kn, L, tid);
        publish_mask(kn, cur ^ 1, pt);
      }
    }
    const float ik[4] = {ik4.x, ik4.y, ik4.z, ik4.w}, iv[4] = {iv4.x, iv4.y, iv4.z, iv4.w};
    float cu[4], ug[4], wk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      cu[j] = cq * ik[j];      // accumulator -> log2-unit score
      ug[j] = gq * iv[j];      // accumulator -> dP (with the dropout scale)
      wk[j] = scale * ik[j];   // dS -> dS / (K group scale): the operand of the product with the SCALED K^T
    }
    if (p_drop > 0.f) attn_drop_keys_in_rows(dk_, q_part, k0, lh, dp);  // dropped probabilities carry no gradient: dP = 0 there
    float wmax = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = r >> 2;
      // (the mask goes into the ARGUMENT, exp2(-inf) = 0: 0 or -inf per key, added to -lse)
      const float4 b4 = *reinterpret_cast<const float4 *>(&sBias[cur][part][8 * j + 4 * lh]);
      const float nb = ((r & 3) == 0 ? b4.x : (r & 3) == 1 ? b4.y : (r & 3) == 2 ? b4.z : b4.w) - my_lse2;
      const float p = __builtin_amdgcn_exp2f(fmaf(s[r], cu[j], nb));
      const float g = dp[r] * ug[j];
      s[r] = p * (g - my_delta) * wk[j];
      wmax = fmaxf(wmax, fabsf(s[r]));
    }
    wmax = fmaxf(wmax, __shfl_xor(wmax, 32, 64));
    const float ratio = online_scale(wmax, bscale);
    if (__builtin_amdgcn_ballot_w64(ratio != 1.f)) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = dq[t][r] * ratio;
          asm volatile("" : "+v"(v));
          dq[t][r] = v;
        }
    }
    // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float x[8] = {s[8 * m], s[8 * m + 1], s[8 * m + 2], s[8 * m + 3], s[8 * m + 4], s[8 * m + 5], s[8 * m + 6], s[8 * m + 7]};
      f16x8 df[2];
      split8g(x, bscale, bscale, df);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f16x8 kt_[2];
        Tile2::frag_cols(sK, 16 * m, 32 * t, lane, kt_);
        dq[t] = mfma3(kt_, df, dq[t]);
      }
    }
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt) {
      stK[pt] = nxK[pt];
      stV[pt] = nxV[pt];
    }
    __syncthreads();
  }
  float un = inv_pow2(bscale);
  if (PARTS == 2) {  // dQ of the second key half, brought to its true scale, is added to the first half's
    float *xch = reinterpret_cast<float *>(smem) + grp * (NT * 16) * 64 + lane;
    if (part == 1) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(t * 16 + r) * 64] = dq[t][r] * un;
    }
    __syncthreads();
    if (part == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[t][r] = dq[t][r] * un + xch[(t * 16 + r) * 64];
      un = 1.f;
    }
  }
  const bool writer = part == 0;
  if (q_ok && writer) {
    float *op = dqkv + (size_t)(b * L + q) * D3 + h * DK;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = t * 32 + 8 * g + 4 * lh;
        *reinterpret_cast<float4 *>(op + d) =
            make_float4(dq[t][4 * g] * un, dq[t][4 * g + 1] * un, dq[t][4 * g + 2] * un, dq[t][4 * g + 3] * un);
      }
  }
  if (row_scale) {
    float am = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) am = fmaxf(am, fabsf(dq[t][r]));
    publish_row_scale(am * un, q_ok && writer, b * L + q, tid, row_scale, row_min, &sMin);
  }
}

// dK, dV: one workgroup = 256 keys of one (protein, head); lane column = key.  The scaled K and V rows of a lane's key
// stay in registers as B operands; Q and dO tiles of 32 queries stream through LDS and serve both as row fragments
// (S = Q K^T, dP = dO V^T) and as transposed fragments (dK^T += Q^T dS, dV^T += dO^T Pd).
// BITS: the dropout decisions are the forward kernel's (keep_bits, one word per key and 32-query tile) - the generator is
// not compiled in (as in attn_bwd_fused_f16x2_kernel below)
template <int DK, int NW, int PARTS, bool BITS>
__global__ __launch_bounds__(64 * NW, 1) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dkv_f16x2_kernel(
    const float *__restrict__ qkv, const int64_t *__restrict__ seq, const float *__restrict__ d_o,
    const float *__restrict__ lse, const float *__restrict__ delta, int L, int H, float p_drop, uint64_t seed,
    uint32_t stream_id, float *__restrict__ dqkv, uint32_t *__restrict__ row_scale, uint32_t *__restrict__ row_min,
    const uint32_t *__restrict__ keep_bits) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  __shared__ __attribute__((aligned(16))) float sLse[2][PARTS][TR], sDel[2][PARTS][TR];
  __shared__ __attribute__((aligned(16))) float sInvQ[2][PARTS][8], sInvG[2][PARTS][8];
  __shared__ unsigned int sMin;
  constexpr int NG = NW / PARTS;  // key groups of the workgroup
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int grp = PARTS == 1 ? wave : wave % NG, part = PARTS == 1 ? 0 : wave / NG;
  const int b = blockIdx.z, h = blockIdx.y, key0 = blockIdx.x * (32 * NG) + grp * 32;
  const int D = H * DK, D3 = 3 * D;
  const float *base = qkv + (size_t)b * L * D3 + h * DK;
  const float *gbase = d_o + (size_t)b * L * D + h * DK;
  const float *lse_b = lse + ((size_t)b * H + h) * L, *del_b = delta + ((size_t)b * H + h) * L;
  const int key = key0 + l31;
  const bool k_ok = key < L;
  const bool k_valid = k_ok && seq[(size_t)b * L + key] != PTAMD_PAD_ID;
  constexpr int KS = DK / 16, NT = DK / 32;
  const float scale = DK == 64 ? 0.125f : 0.17677669529663687f;
  const int lkb = (L + 31) & ~31;   // keys (and query tiles x 32) of the keep_bits layout
  const AttnDrop dk_ = make_attn_drop(seed, stream_id, (uint32_t)(b * H + h), p_drop);
  const float ks = p_drop > 0.f ? dk_.ks : 1.f;

  f16x8 kf[KS][2], vf[KS][2];
  const float ikl = load_row_scaled<KS>(base + D, D3, min(key, L - 1), k_ok, lh, kf);
  const float ivl = load_row_scaled<KS>(base + 2 * D, D3, min(key, L - 1), k_ok, lh, vf);
  const float ck = scale * LOG2E * ikl, gk = ivl * ks;

  f32x16 dk[NT], dv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[t][r] = dv[t][r] = 0.f;
  float bscale = BSCALE0;  // common power of two of the dS operand (per key = per accumulator column)
  float g_run = 0.f;       // largest inverse dO group scale so far (wavefront-uniform)

  Stage<DK, NW> stQ[PARTS], stG[PARTS];
  const int ntiles = ((L + TR - 1) / TR + PARTS - 1) / PARTS;  // query tiles of one part: part p walks tiles p ntiles ..
  auto tile = [&](int buf, int pt) __attribute__((always_inline)) { return smem + (buf * PARTS + pt) * BUF; };
  float r_lse[PARTS], r_del[PARTS];
  TileRows<DK, NW> rows_q[PARTS], rows_g[PARTS];
  Stage<DK, NW> nxQ[PARTS], nxG[PARTS];
#pragma unroll
  for (int pt = 0; pt < PARTS; ++pt) {
    const int first = pt * ntiles * TR;
    rows_q[pt].init(D3, L, tid, first);
    rows_g[pt].init(D, L, tid, first);
    stQ[pt].load(base, rows_q[pt].next());
    stG[pt].load(gbase, rows_g[pt].next());
    stQ[pt].store(tile(0, pt), sInvQ[0][pt], first, L, tid);
    stG[pt].store(tile(0, pt) + Tile2::ELEMS, sInvG[0][pt], first, L, tid);
    r_lse[pt] = r_del[pt] = 0.f;
    if (tid < TR) {
      sLse[0][pt][tid] = first + tid < L ? lse_b[first + tid] * LOG2E : INFINITY;
      sDel[0][pt][tid] = first + tid < L ? del_b[first + tid] : 0.f;
    }
    stQ[pt].load(base, rows_q[pt].next());
    stG[pt].load(gbase, rows_g[pt].next());
  }
  __syncthreads();

  for (int qt = 0; qt < ntiles; ++qt) {
    const int qq0 = (part * ntiles + qt) * TR, cur = qt & 1;
    const bool more = qt + 1 < ntiles;
    const unsigned short *sQ = tile(cur, part), *sG = sQ + Tile2::ELEMS;
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt) {
      nxQ[pt].load(base, rows_q[pt].next());
      nxG[pt].load(gbase, rows_g[pt].next());
      if (more && tid < TR) {
        const int qn = (pt * ntiles + qt + 1) * TR + tid;
        r_lse[pt] = qn < L ? lse_b[qn] * LOG2E : INFINITY;
        r_del[pt] = qn < L ? del_b[qn] : 0.f;
      }
    }
    // (BITS) this lane's key against the 32 queries of the tile, bit = query: requested here, used behind the products
    uint32_t kword = 0xffffffffu;
    if (BITS && qq0 < lkb) kword = keep_bits[((size_t)(b * H + h) * (lkb >> 5) + (qq0 >> 5)) * lkb + min(key, lkb - 1)];
    const float4 iq4 = *reinterpret_cast<const float4 *>(&sInvQ[cur][part][4 * lh]);
    const float4 iga = *reinterpret_cast<const float4 *>(&sInvG[cur][part][0]), igb = *reinterpret_cast<const float4 *>(&sInvG[cur][part][4]);
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int st = 0; st < KS; ++st) {
      f16x8 qa[2], ga[2];
      Tile2::frag_rows(sQ, st, lane, qa);
      Tile2::frag_rows(sG, st, lane, ga);
      s = mfma3(qa, kf[st], s);     // S[q][key]
      dp = mfma3(ga, vf[st], dp);   // dP[q][key] = dO V^T
    }
    if (more) {
#pragma unroll
      for (int pt = 0; pt < PARTS; ++pt) {
        const int qn = (pt * ntiles + qt + 1) * TR;
        stQ[pt].store(tile(cur ^ 1, pt), sInvQ[cur ^ 1][pt], qn, L, tid);
        stG[pt].store(tile(cur ^ 1, pt) + Tile2::ELEMS, sInvG[cur ^ 1][pt], qn, L, tid);
        if (tid < TR) {
          sLse[cur ^ 1][pt][tid] = r_lse[pt];
          sDel[cur ^ 1][pt][tid] = r_del[pt];
        }
      }
    }
    // common scale of the dO groups: dV accumulates (2^14 / g_run) sum Pd dO
    const float gt = fmaxf(fmaxf(fmaxf(iga.x, iga.y), fmaxf(iga.z, iga.w)), fmaxf(fmaxf(igb.x, igb.y), fmaxf(igb.z, igb.w)));
    if (gt > g_run) {  // (wavefront-uniform)
      const float resc = g_run * inv_pow2(gt);
      g_run = gt;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = dv[t][r] * resc;
          asm volatile("" : "+v"(v));
          dv[t][r] = v;
        }
    }
    const float gn = inv_pow2(g_run);  // (2^127 while every dO group seen so far is zero: multiplied LAST, after 0 * 2^14)
    const float4 igh = lh ? igb : iga;
    const float iq[4] = {iq4.x, iq4.y, iq4.z, iq4.w}, ig[4] = {igh.x, igh.y, igh.z, igh.w};
    float cu[4], ug[4], wq[4], fp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      cu[j] = ck * iq[j];
      ug[j] = gk * ig[j];
      wq[j] = scale * iq[j];
      fp[j] = ig[j] * TWO14 * gn;  // <= 2^14
    }
    f32x16 pd;  // dropped probabilities (operand of dV), without the 1 / (1 - p): that is applied to dV at the end
    // bit (r & 3) + 8 (r >> 2) of `keepw` = register r is kept: the stored word shifted by the lane half, or the generator's 16
    // bits spread to the same positions
    uint32_t keepw;
    if (BITS) {
      keepw = kword >> (4 * lh);
    } else {
      const uint32_t kb16 = p_drop > 0.f ? attn_keep_bits_queries_in_rows_paired(dk_, (uint32_t)key, qq0, lh) : 0xffffu;
      keepw = (kb16 & 0xfu) | ((kb16 & 0xf0u) << 4) | ((kb16 & 0xf00u) << 8) | ((kb16 & 0xf000u) << 12);
    }
    float wmax = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = r >> 2;
      // (LDS broadcast reads, one float4 per register quadruple: registers are the scarce resource here; query rows beyond
      // L carry lse = +inf, so the row mask costs nothing per element and the key mask is a loop-invariant select)
      const float4 l4 = *reinterpret_cast<const float4 *>(&sLse[cur][part][8 * j + 4 * lh]);
      const float4 d4 = *reinterpret_cast<const float4 *>(&sDel[cur][part][8 * j + 4 * lh]);
      const float my_l = (r & 3) == 0 ? l4.x : (r & 3) == 1 ? l4.y : (r & 3) == 2 ? l4.z : l4.w;
      const float my_d = (r & 3) == 0 ? d4.x : (r & 3) == 1 ? d4.y : (r & 3) == 2 ? d4.z : d4.w;
      const float p = __builtin_amdgcn_exp2f(k_valid ? fmaf(s[r], cu[j], -my_l) : -INFINITY);
      float g = dp[r] * ug[j], pk = p;
      if (p_drop > 0.f) {
        g = keep_or_zero(g, keepw, (r & 3) + 8 * (r >> 2));
        pk = keep_or_zero(p, keepw, (r & 3) + 8 * (r >> 2));
      }
      pd[r] = pk;
      s[r] = p * (g - my_d) * wq[j];  // dS[q][key] / (Q group scale)
      wmax = fmaxf(wmax, fabsf(s[r]));
    }
    wmax = fmaxf(wmax, __shfl_xor(wmax, 32, 64));
    const float ratio = online_scale(wmax, bscale);
    if (__builtin_amdgcn_ballot_w64(ratio != 1.f)) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = dk[t][r] * ratio;
          asm volatile("" : "+v"(v));
          dk[t][r] = v;
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float xs[8] = {s[8 * m], s[8 * m + 1], s[8 * m + 2], s[8 * m + 3], s[8 * m + 4], s[8 * m + 5], s[8 * m + 6], s[8 * m + 7]};
      const float xp[8] = {pd[8 * m], pd[8 * m + 1], pd[8 * m + 2], pd[8 * m + 3], pd[8 * m + 4], pd[8 * m + 5], pd[8 * m + 6], pd[8 * m + 7]};
      f16x8 dsf[2], pdf[2];
      split8g(xs, bscale, bscale, dsf);
      split8g(xp, fp[2 * m], fp[2 * m + 1], pdf);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f16x8 qt_[2], gt_[2];
        Tile2::frag_cols(sQ, 16 * m, 32 * t, lane, qt_);
        dk[t] = mfma3(qt_, dsf, dk[t]);   // dK^T[d][key] += Q^T dS
        Tile2::frag_cols(sG, 16 * m, 32 * t, lane, gt_);
        dv[t] = mfma3(gt_, pdf, dv[t]);   // dV^T[d][key] += dO^T Pd
      }
    }
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt) {
      stQ[pt] = nxQ[pt];
      stG[pt] = nxG[pt];
    }
    __syncthreads();
  }
  float uk = inv_pow2(bscale), uv = ks * g_run * INV_TWO14;
  if (PARTS == 2) {  // dK, dV of the second query half, brought to their true scales, are added to the first half's
    float *xch = reinterpret_cast<float *>(smem) + grp * (2 * NT * 16) * 64 + lane;
    if (part == 1) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          xch[(t * 16 + r) * 64] = dk[t][r] * uk;
          xch[(NT * 16 + t * 16 + r) * 64] = dv[t][r] * uv;
        }
    }
    __syncthreads();
    if (part == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          dk[t][r] = dk[t][r] * uk + xch[(t * 16 + r) * 64];
          dv[t][r] = dv[t][r] * uv + xch[(NT * 16 + t * 16 + r) * 64];
        }
      uk = uv = 1.f;
    }
  }
  const bool writer = part == 0;
  if (row_scale) {
    float ak = 0.f, av = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        ak = fmaxf(ak, fabsf(dk[t][r]));
        av = fmaxf(av, fabsf(dv[t][r]));
      }
    publish_row_scale(fmaxf(ak * uk, av * uv), k_ok && writer, b * L + key, tid, row_scale, row_min, &sMin);
  }
  if (k_ok && writer) {
    float *okp = dqkv + (size_t)(b * L + key) * D3 + D + h * DK, *ovp = okp + D;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = t * 32 + 8 * g + 4 * lh;
        *reinterpret_cast<float4 *>(okp + d) =
            make_float4(dk[t][4 * g] * uk, dk[t][4 * g + 1] * uk, dk[t][4 * g + 2] * uk, dk[t][4 * g + 3] * uk);
        *reinterpret_cast<float4 *>(ovp + d) =
            make_float4(dv[t][4 * g] * uv, dv[t][4 * g + 1] * uv, dv[t][4 * g + 2] * uv, dv[t][4 * g + 3] * uv);
      }
  }
}

// =================================================================================================== fused backward
// dQ, dK and dV in ONE sweep (round 4; dk = 64, 8 wavefronts; launched when protein x head workgroups fill the chip).
// The reference's backward of softmax(QK^T / sqrt(dk)) V (Attention.py:14-22) is one pass over the score matrix; the two
// kernels above walk it twice (exp2 / dropout hash / two-term splitting in both).  This kernel is the dK/dV kernel - one
// workgroup per (protein, head), a wavefront per 32 keys, query tiles streaming through LDS - with two additions:
//   * an OUTER loop over the 256-key blocks of the protein, so that every contribution to a query's dQ comes from this
//     workgroup, in a fixed order (first block: store, later blocks: read - add - store by the same lane; no atomics, no
//     slabs, bit-reproducible);
//   * dQ^T[d][q] += K^T[d][key] dS^T[key][q] for a query tile over the 256 keys of the block: every wavefront leaves its dS
//     tile (keys in lanes) in LDS as two f16 planes [key][q], scaled by the lane's inverse K row scale and ONE power of two
//     per wavefront and tile (published beside it); the block's scaled K rows sit in LDS as planes [key][d] for the whole
//     block; behind a barrier wavefront w computes the 16 (d) x 16 (q) piece (w & 3, w >> 2) of dQ^T with
//     v_mfma_f32_16x16x32_f16 over the eight 32-key steps - both operands by transposing reads (ds_read_b64_tr_b16), each
//     step's three products into a fresh accumulator that is added with the step's inverse scale - so no partial dQ is
//     ever exchanged between wavefronts.
// LDS: 48 KB staging (as above) + 64 KB K planes + 32 KB dS planes = 144 KB.  delta comes from attn_delta_kernel.
constexpr int FK = 256;                              // keys of a block = 8 wavefronts x 32
constexpr int KP_PLANE = FK * 64, DS_PLANE = FK * 32;  // f16 elements of a K plane / a dS plane
constexpr size_t FUSED_LDS = ATTN_LDS + (size_t)(2 * KP_PLANE + 2 * DS_PLANE) * sizeof(unsigned short);
// K planes [key][64]: 16-byte chunk c of row r at c ^ (2 ((r >> 1) & 3)) - the transposing reads of a 16 x 16 piece
// (4 rows x 32 bytes per 16 lanes, rows 4 g .. 4 g + 3 of lane group g) then touch every bank once per 32 lanes
__device__ __forceinline__ int kp_off(int row, int d) { return row * 64 + ((((d >> 3) ^ (2 * ((row >> 1) & 3))) & 7) << 3) + (d & 7); }
// dS planes [key][32]: the 32-byte half (q >> 4) of row r at half ^ ((r >> 2) & 1)
__device__ __forceinline__ int ds_off(int row, int q) { return row * 32 + ((((q >> 4) ^ (row >> 2)) & 1) << 4) + (q & 15); }
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void attn_delta_kernel(const float *__restrict__ o_fwd, const float *__restrict__ d_o,
                                                        int rows, int L, int H, int dk, float *__restrict__ delta) {
  // delta[b, h, q] = sum_d dO[b q, h dk + d] O[b q, h dk + d]: dk / 4 lanes per (row, head), one float4 each
  const int lp = dk >> 2, per_row = H * lp;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t row = i / per_row;
  const int c = (int)(i % per_row), h = c / lp;
  float acc = 0.f;
  if (row < (size_t)rows) {
    const float4 g = *reinterpret_cast<const float4 *>(d_o + row * (size_t)(H * dk) + 4 * c);
    const float4 o = *reinterpret_cast<const float4 *>(o_fwd + row * (size_t)(H * dk) + 4 * c);
    acc = (g.x * o.x + g.y * o.y) + (g.z * o.z + g.w * o.w);
  }
  for (int off = lp >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);   // (lp lanes of a head are adjacent, lp <= 16)
  if (row < (size_t)rows && (c % lp) == 0) {
    const size_t b = row / L, q = row % L;
    delta[(b * H + h) * L + q] = acc;
  }
}

// BITS: the dropout decisions come from the forward kernel (keep_bits) - the generator is not compiled in.
// KVP: the key block's K and V rows come PRE-SPLIT (kv_format.h, written by the QKV product's epilogue) - a row's scale is
// then that of its group of four tokens instead of its own; nothing else changes (Q and dO tiles are staged as before).
// SPLIT (round 6: few (protein, head) pairs - the per-GPU share of a strongly scaled batch): the same sweep cut into MORE
// workgroups.  1: one workgroup per (pair, 256-key block) - dK / dV of the block are complete as before, the block's
// contribution to dQ goes to slab `kb` of `dq_part` ([key blocks][tokens][D]) instead of read-add-store.  2: the query tiles
// of a key block are cut into `qs` ranges as well, one workgroup each - its dK / dV (of its query range) go to slab `qsx` of
// `dkv_part` ([ranges][tokens][2 D]).  attn_bwd_split_reduce_kernel sums the slabs in a fixed order (dQ: key-block order, the
// bits of the unsplit sweep), writes dqkv and takes over the row scales of what it sums.
template <bool BITS, bool KVP = false, int SPLIT = 0>
__global__ __launch_bounds__(512, 1) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_fused_f16x2_kernel(
    const float *__restrict__ qkv, const int64_t *__restrict__ seq, const float *__restrict__ d_o,
    const float *__restrict__ lse, const float *__restrict__ delta, int L, int H, float p_drop, uint64_t seed,
    uint32_t stream_id, float *__restrict__ dqkv, uint32_t *__restrict__ row_scale, uint32_t *__restrict__ row_min,
    const uint32_t *__restrict__ keep_bits, const char *__restrict__ kvp = nullptr, const float *__restrict__ kv_inv = nullptr,
    int kv_nt = 0, float *__restrict__ dq_part = nullptr, float *__restrict__ dkv_part = nullptr, int qs = 1) {
  constexpr int DK = 64, NW = 8, KS = DK / 16, NT = DK / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  __shared__ __attribute__((aligned(16))) float sLse[2][TR], sDel[2][TR];
  __shared__ __attribute__((aligned(16))) float sInvQ[2][8], sInvG[2][8];
  __shared__ float sInvC[NW];
  __shared__ unsigned int sMin;
  unsigned short *const sKP = smem + 2 * BUF, *const sDS = sKP + 2 * KP_PLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int i16 = lane & 15, g16 = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int D = H * DK, D3 = 3 * D;
  const float *base = qkv + (size_t)b * L * D3 + h * DK;
  const float *gbase = d_o + (size_t)b * L * D + h * DK;
  const float *lse_b = lse + ((size_t)b * H + h) * L, *del_b = delta + ((size_t)b * H + h) * L;
  const float scale = 0.125f;
  const AttnDrop dk_ = make_attn_drop(seed, stream_id, (uint32_t)(b * H + h), p_drop);
  const float ks = p_drop > 0.f ? dk_.ks : 1.f;
  const int ntiles = (L + TR - 1) / TR, nkb = (L + FK - 1) / FK;
  const int lk = (L + 31) & ~31;
  // the key blocks and query tiles of THIS workgroup (everything unless SPLIT)
  int kb_first = 0, kb_last = nkb, qt_first = 0, qt_last = ntiles, qsx = 0;
  if (SPLIT) {
    kb_first = (int)blockIdx.x / qs;
    kb_last = kb_first + 1;
    if (SPLIT == 2) {
      const int per = (ntiles + qs - 1) / qs;
      qsx = (int)blockIdx.x % qs;
      qt_first = qsx * per;
      qt_last = min(ntiles, qt_first + per);
    }
  }
  const size_t T_all = (size_t)gridDim.z * L;   // tokens of the batch (slab stride)
  const bool use_bits = BITS && p_drop > 0.f;   // (uniform)
  const int db = 16 * (wave & 3), qb = 16 * (wave >> 2);   // this wavefront's piece of a tile's dQ^T
  auto tile = [&](int buf) __attribute__((always_inline)) { return smem + buf * BUF; };
  uint32_t my_min = 0x7F000000u;                          // smallest f16x2 scale of the dQ rows this thread published
  // this lane's transposing reads of a 32-key step st: rows 32 st + 4 g16 + (i16 >> 2) (+ 16), four columns from db / qb +
  // 4 (i16 & 3); the swizzles depend on the row only through bits that 32 st and 16 leave alone, so a step is + 32 rows
  // this lane's own key row in the K planes: chunk c = 2 st + lh of row r sits at c ^ (2 ((r >> 1) & 3))
  const int kswz = (2 * (((wave * 32 + l31) >> 1) & 3)) ^ lh;
  const unsigned short *const kself = sKP + (wave * 32 + l31) * 64;
  const unsigned short *const ka0 = sKP + kp_off(4 * g16 + (i16 >> 2), db + 4 * (i16 & 3));
  const unsigned short *const da0 = sDS + ds_off(4 * g16 + (i16 >> 2), qb + 4 * (i16 & 3));

  for (int kb = kb_first; kb < kb_last; ++kb) {
    const int key = kb * FK + wave * 32 + l31;
    const bool k_ok = key < L;
    const bool k_valid = k_ok && seq[(size_t)b * L + key] != PTAMD_PAD_ID;
    const uint32_t *const keep_row = keep_bits + (size_t)(b * H + h) * (lk >> 5) * lk + min(key, lk - 1);
    f16x8 vf[KS][2];
    float ikl;
    {  // the scaled K rows of the block go to LDS as planes and are read from there by BOTH products that need them (S = Q K^T
       // as row fragments, dQ^T += K^T dS^T by transposing reads): 32 registers less than holding them (the dK/dV kernel's 256
       // are all taken).  (The previous block's readers are behind the last barrier of its loop.)
      f16x8 kf[KS][2];
      if (KVP) ikl = load_row_planes<KS>(kvp, kv_inv, ptkv::tile_index(0, h, 0, H, kv_nt), b * L + min(key, L - 1), k_ok, lh, kf);
      else ikl = load_row_scaled<KS>(base + D, D3, min(key, L - 1), k_ok, lh, kf);
      const int row = wave * 32 + l31;
#pragma unroll
      for (int st = 0; st < KS; ++st)
#pragma unroll
        for (int t = 0; t < 2; ++t) *reinterpret_cast<f16x8 *>(sKP + t * KP_PLANE + kp_off(row, 16 * st + 8 * lh)) = kf[st][t];
    }
    const float ivl = KVP ? load_row_planes<KS>(kvp, kv_inv, ptkv::tile_index(1, h, 0, H, kv_nt), b * L + min(key, L - 1), k_ok, lh, vf)
                          : load_row_scaled<KS>(base + 2 * D, D3, min(key, L - 1), k_ok, lh, vf);
    const float ck = scale * LOG2E * ikl, gk = ivl * ks;
    f32x16 dk[NT], dv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) dk[t][r] = dv[t][r] = 0.f;
    float bscale = BSCALE0, g_run = 0.f;

    Stage<DK, NW> stQ, stG, nxQ, nxG;
    TileRows<DK, NW> rows_q, rows_g;
    float r_lse = 0.f, r_del = 0.f;
    rows_q.init(D3, L, tid, qt_first * TR);
    rows_g.init(D, L, tid, qt_first * TR);
    stQ.load(base, rows_q.next());
    stG.load(gbase, rows_g.next());
    stQ.store(tile(0), sInvQ[0], qt_first * TR, L, tid);
    stG.store(tile(0) + Tile2::ELEMS, sInvG[0], qt_first * TR, L, tid);
    if (tid < TR) {
      const int q0 = qt_first * TR + tid;
      sLse[0][tid] = q0 < L ? lse_b[q0] * LOG2E : INFINITY;
      sDel[0][tid] = q0 < L ? del_b[q0] : 0.f;
    }
    stQ.load(base, rows_q.next());
    stG.load(gbase, rows_g.next());
    __syncthreads();

    for (int qt = qt_first; qt < qt_last; ++qt) {
      const int qq0 = qt * TR, cur = (qt - qt_first) & 1;
      const bool more = qt + 1 < qt_last;
      const unsigned short *sQ = tile(cur), *sG = sQ + Tile2::ELEMS;
      nxQ.load(base, rows_q.next());
      nxG.load(gbase, rows_g.next());
      if (more && tid < TR) {
        const int qn = (qt + 1) * TR + tid;
        r_lse = qn < L ? lse_b[qn] * LOG2E : INFINITY;
        r_del = qn < L ? del_b[qn] : 0.f;
      }
      // the forward kernel's dropout decisions of this lane's key against the 32 queries of the tile (bit = query), if the
      // caller kept them: requested here, used behind the two products
      uint32_t kword = 0xffffffffu;
      if (use_bits) kword = keep_row[(size_t)qt * lk];
      const float4 iq4 = *reinterpret_cast<const float4 *>(&sInvQ[cur][4 * lh]);
      const float4 iga = *reinterpret_cast<const float4 *>(&sInvG[cur][0]), igb = *reinterpret_cast<const float4 *>(&sInvG[cur][4]);
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
      for (int st = 0; st < KS; ++st) {
        f16x8 qa[2], ga[2];
        Tile2::frag_rows(sQ, st, lane, qa);
        Tile2::frag_rows(sG, st, lane, ga);
        f16x8 kfr[2];   // this lane's key row, d = 16 st + 8 lh + 0..7, from the K planes
#pragma unroll
        for (int t = 0; t < 2; ++t) kfr[t] = *reinterpret_cast<const f16x8 *>(kself + t * KP_PLANE + ((((2 * st) ^ kswz) & 7) << 3));
        s = mfma3(qa, kfr, s);     // S[q][key]
        dp = mfma3(ga, vf[st], dp);   // dP[q][key] = dO V^T
      }
      if (more) {
        const int qn = (qt + 1) * TR;
        stQ.store(tile(cur ^ 1), sInvQ[cur ^ 1], qn, L, tid);
        stG.store(tile(cur ^ 1) + Tile2::ELEMS, sInvG[cur ^ 1], qn, L, tid);
        if (tid < TR) {
          sLse[cur ^ 1][tid] = r_lse;
          sDel[cur ^ 1][tid] = r_del;
        }
      }
      const float gt = fmaxf(fmaxf(fmaxf(iga.x, iga.y), fmaxf(iga.z, iga.w)), fmaxf(fmaxf(igb.x, igb.y), fmaxf(igb.z, igb.w)));
      if (gt > g_run) {  // (wavefront-uniform)
        const float resc = g_run * inv_pow2(gt);
        g_run = gt;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = dv[t][r] * resc;
            asm volatile("" : "+v"(v));
            dv[t][r] = v;
          }
      }
      const float gn = inv_pow2(g_run);
      const float4 igh = lh ? igb : iga;
      const float iq[4] = {iq4.x, iq4.y, iq4.z, iq4.w}, ig[4] = {igh.x, igh.y, igh.z, igh.w};
      float cu[4], ug[4], wq[4], fp[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        cu[j] = ck * iq[j];
        ug[j] = gk * ig[j];
        wq[j] = scale * iq[j];
        fp[j] = ig[j] * TWO14 * gn;
      }
      f32x16 pd;   // dropped probabilities (operand of dV)
      // bit (r & 3) + 8 (r >> 2) of `keepw` = register r (query qq0 + 4 lh + (r & 3) + 8 (r >> 2)) is kept: the stored word
      // shifted by the lane half, or the generator's 16 bits spread to the same positions
      uint32_t keepw;
      if (BITS) {
        keepw = kword >> (4 * lh);
      } else {
        const uint32_t kb16 = p_drop > 0.f ? attn_keep_bits_queries_in_rows_paired(dk_, (uint32_t)key, qq0, lh) : 0xffffu;
        keepw = (kb16 & 0xfu) | ((kb16 & 0xf0u) << 4) | ((kb16 & 0xf00u) << 8) | ((kb16 & 0xf000u) << 12);
      }
      float wmax = 0.f;
      float gm[4] = {0.f, 0.f, 0.f, 0.f};   // max |dS / (Q group scale)| per register quadruple (= per Q group)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = r >> 2;
        const float4 l4 = *reinterpret_cast<const float4 *>(&sLse[cur][8 * j + 4 * lh]);
        const float4 d4 = *reinterpret_cast<const float4 *>(&sDel[cur][8 * j + 4 * lh]);
        const float my_l = (r & 3) == 0 ? l4.x : (r & 3) == 1 ? l4.y : (r & 3) == 2 ? l4.z : l4.w;
        const float my_d = (r & 3) == 0 ? d4.x : (r & 3) == 1 ? d4.y : (r & 3) == 2 ? d4.z : d4.w;
        const float p = __builtin_amdgcn_exp2f(k_valid ? fmaf(s[r], cu[j], -my_l) : -INFINITY);
        float g = dp[r] * ug[j], pk = p;
        if (p_drop > 0.f) {
          g = keep_or_zero(g, keepw, (r & 3) + 8 * (r >> 2));
          pk = keep_or_zero(p, keepw, (r & 3) + 8 * (r >> 2));
        }
        pd[r] = pk;
        s[r] = p * (g - my_d) * wq[j];   // dS[q][key] / (Q group scale)
        gm[j] = fmaxf(gm[j], fabsf(s[r]));
      }
      // ---- this wavefront's dS tile into LDS: planes [key][q] of dS scale / (K row scale) = s (Q group scale) / (K row
      // scale) - powers of two, exact - times one power of two for the whole tile of the wavefront
      {
        float fq[4];
        uint32_t w2 = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // (a register quadruple that is all zero - masked keys, query rows beyond L, dO = 0 - takes the factor 0: its Q
          // group scale may be 2^127 and the wavefront's scale below 2^127, and 0 * inf is not 0)
          fq[j] = gm[j] > 0.f ? ikl * inv_pow2(iq[j]) : 0.f;
          w2 = max(w2, abs_bits(gm[j] * fq[j]));
          wmax = fmaxf(wmax, gm[j]);
        }
        const uint32_t sbits = pt_row_scale_bits(group_umax<64>(w2));
        const float cw = __uint_as_float(sbits);
        const int row = wave * 32 + l31;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          uint2 h1, h2;
          const float cj = fminf(cw * fq[j], 8.507059173023462e37f);   // 2^126: finite whatever the magnitudes (see above)
          split_quad_f16(s[4 * j], s[4 * j + 1], s[4 * j + 2], s[4 * j + 3], cj, cj, cj, cj, h1, h2);
          const int off = ds_off(row, 8 * j + 4 * lh);
          *reinterpret_cast<uint2 *>(sDS + off) = h1;
          *reinterpret_cast<uint2 *>(sDS + DS_PLANE + off) = h2;
        }
        if (lane == 0) sInvC[wave] = __uint_as_float((254u << 23) - sbits);
      }
      wmax = fmaxf(wmax, __shfl_xor(wmax, 32, 64));
      const float ratio = online_scale(wmax, bscale);
      if (__builtin_amdgcn_ballot_w64(ratio != 1.f)) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = dk[t][r] * ratio;
            asm volatile("" : "+v"(v));
            dk[t][r] = v;
          }
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const float xs[8] = {s[8 * m], s[8 * m + 1], s[8 * m + 2], s[8 * m + 3], s[8 * m + 4], s[8 * m + 5], s[8 * m + 6], s[8 * m + 7]};
        const float xp[8] = {pd[8 * m], pd[8 * m + 1], pd[8 * m + 2], pd[8 * m + 3], pd[8 * m + 4], pd[8 * m + 5], pd[8 * m + 6], pd[8 * m + 7]};
        f16x8 dsf[2], pdf[2];
        split8g(xs, bscale, bscale, dsf);
        split8g(xp, fp[2 * m], fp[2 * m + 1], pdf);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          f16x8 qt_[2], gt_[2];
          Tile2::frag_cols(sQ, 16 * m, 32 * t, lane, qt_);
          dk[t] = mfma3(qt_, dsf, dk[t]);   // dK^T[d][key] += Q^T dS
          Tile2::frag_cols(sG, 16 * m, 32 * t, lane, gt_);
          dv[t] = mfma3(gt_, pdf, dv[t]);   // dV^T[d][key] += dO^T Pd
        }
      }
      stQ = nxQ;
      stG = nxG;
      __syncthreads();  // every wavefront's dS tile (and the next staged tiles) is in LDS; nobody reads the current tiles any more
      // ---- dQ^T[db .. db + 16][qb .. qb + 16] of this query tile over the 256 keys of the block
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      // lane (i16, g16): query qq0 + qb + i16, d = db + 4 g16 + 0..3; what the earlier key blocks left there is requested
      // here and added behind the products (it was a dependent load in front of the tile's last barrier)
      const int dq_q = qq0 + qb + i16;
      float *const dq_p = dqkv + (size_t)(b * L + min(dq_q, L - 1)) * D3 + h * DK + db + 4 * g16;
      float4 dq_old = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!SPLIT && kb > 0) dq_old = *reinterpret_cast<const float4 *>(dq_p);
#pragma unroll 2
      for (int st = 0; st < NW; ++st) {   // keys 32 st .. 32 st + 31 = the rows wavefront st wrote; offsets: see ka0 / da0
        f16x8 a[2], bq[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const unsigned short *ka = ka0 + t * KP_PLANE + st * (32 * 64), *da = da0 + t * DS_PLANE + st * (32 * 32);
          const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)ka);
          const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(ka + 16 * 64));
          const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)da);
          const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(da + 16 * 32));
          a[t] = __builtin_bit_cast(f16x8, (s16x8)__builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
          bq[t] = __builtin_bit_cast(f16x8, (s16x8)__builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], bq[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], bq[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], bq[0], c, 0, 0, 0);
        const float ic = sInvC[st];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(c[e], ic, acc[e]);
      }
      {
        const int q = dq_q;
        float *op = dq_p;
        if (SPLIT)   // this key block's contribution: slab kb, [token][D] (summed by attn_bwd_split_reduce_kernel)
          op = dq_part + ((size_t)kb * T_all + (size_t)b * L + min(q, L - 1)) * D + h * DK + db + 4 * g16;
        float4 v = make_float4(acc[0], acc[1], acc[2], acc[3]);
        if (!SPLIT && kb > 0 && q < L) {
          v.x += dq_old.x; v.y += dq_old.y; v.z += dq_old.z; v.w += dq_old.w;
        }
        if (q < L) *reinterpret_cast<float4 *>(op) = v;
        if (!SPLIT && row_scale && kb == nkb - 1) {   // the row's f16x2 scale: the four lane groups hold 16 d of the row's 64
          uint32_t am = q < L ? umax4(v) : 0u;
          am = max(am, (uint32_t)__shfl_xor((int)am, 16, 64));
          am = max(am, (uint32_t)__shfl_xor((int)am, 32, 64));
          const uint32_t sb = pt_row_scale_bits(am);
          if (q < L && g16 == 0) atomicMin(row_scale + (size_t)b * L + q, sb);
          if (q < L) my_min = min(my_min, sb);
        }
      }
      __syncthreads();  // the dS planes (and, behind the last tile of a block, the K planes) may be rewritten
    }
    // ---- dK, dV of this key block
    const float uk = inv_pow2(bscale), uv = ks * g_run * INV_TWO14;
    if (row_scale && SPLIT != 2) {
      float ak = 0.f, av = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          ak = fmaxf(ak, fabsf(dk[t][r]));
          av = fmaxf(av, fabsf(dv[t][r]));
        }
      float amax = fmaxf(ak * uk, av * uv);
      amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
      const uint32_t sb = k_ok ? pt_row_scale_bits(__float_as_uint(amax)) : 0x7F000000u;
      if (k_ok && lane < 32) atomicMin(row_scale + (size_t)b * L + key, sb);
      my_min = min(my_min, sb);
    }
    if (k_ok) {
      float *okp = dqkv + (size_t)(b * L + key) * D3 + D + h * DK, *ovp = okp + D;
      if (SPLIT == 2) {   // the share of this workgroup's query range: slab qsx, [token][2 D]
        okp = dkv_part + ((size_t)qsx * T_all + (size_t)b * L + key) * (2 * D) + h * DK;
        ovp = okp + D;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = t * 32 + 8 * g + 4 * lh;
          *reinterpret_cast<float4 *>(okp + d) =
              make_float4(dk[t][4 * g] * uk, dk[t][4 * g + 1] * uk, dk[t][4 * g + 2] * uk, dk[t][4 * g + 3] * uk);
          *reinterpret_cast<float4 *>(ovp + d) =
              make_float4(dv[t][4 * g] * uv, dv[t][4 * g + 1] * uv, dv[t][4 * g + 2] * uv, dv[t][4 * g + 3] * uv);
        }
    }
  }
  if (row_scale && row_min && SPLIT != 2) {  // the smallest scale of all rows this workgroup wrote: one global atomic per copy
    if (tid == 0) sMin = 0x7F000000u;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) my_min = min(my_min, (uint32_t)__shfl_xor((int)my_min, o, 64));
    if (lane == 0) atomicMin(&sMin, my_min);
    __syncthreads();
    if (tid < 4) atomicMin(row_min + tid, sMin);
  }
}

// The slabs of the split sweep -> dqkv, one wavefront per token: dQ = the key blocks' contributions in block order (the bits
// of the unsplit sweep's read-add-store), and with DKV the query ranges' dK | dV in range order; the f16x2 scale of what was
// summed joins the row's scale (atomicMin: with SPLIT = 1 the sweep itself contributed the K / V columns) and the batch minimum.
template <bool DKV>
__global__ __launch_bounds__(256) void attn_bwd_split_reduce_kernel(const float *__restrict__ dq_part, int nkb,
                                                                  const float *__restrict__ dkv_part, int qs, size_t T, int D,
                                                                  float *__restrict__ dqkv, uint32_t *__restrict__ row_scale,
                                                                  uint32_t *__restrict__ row_min) {
  // two token rows per wavefront, their loads interleaved (one row's few dependent 16-byte loads per lane left the memory
  // system idle: 29.6 us for 50 MB at 16 proteins)
  __shared__ unsigned int sMin;
  const int lane = threadIdx.x & 63;
  const size_t t0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
  if (threadIdx.x == 0) sMin = 0x7F000000u;
  __syncthreads();
  uint32_t sb_min = 0x7F000000u;
  if (t0 < T) {
    const bool two = t0 + 1 < T;
    const size_t t1 = two ? t0 + 1 : t0;
    uint32_t am0 = 0u, am1 = 0u;
    float *out0 = dqkv + t0 * (size_t)(3 * D), *out1 = dqkv + t1 * (size_t)(3 * D);
    for (int c = lane * 4; c < D; c += 256) {
      float4 v = *reinterpret_cast<const float4 *>(dq_part + t0 * D + c), w = *reinterpret_cast<const float4 *>(dq_part + t1 * D + c);
      for (int k = 1; k < nkb; ++k) {
        const float4 o = *reinterpret_cast<const float4 *>(dq_part + ((size_t)k * T + t0) * D + c);
        const float4 q = *reinterpret_cast<const float4 *>(dq_part + ((size_t)k * T + t1) * D + c);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        w.x += q.x; w.y += q.y; w.z += q.z; w.w += q.w;
      }
      *reinterpret_cast<float4 *>(out0 + c) = v;
      if (two) *reinterpret_cast<float4 *>(out1 + c) = w;
      am0 = max(am0, umax4(v));
      am1 = max(am1, umax4(w));
    }
    if (DKV) {
      for (int c = lane * 4; c < 2 * D; c += 256) {
        float4 v = *reinterpret_cast<const float4 *>(dkv_part + t0 * (size_t)(2 * D) + c);
        float4 w = *reinterpret_cast<const float4 *>(dkv_part + t1 * (size_t)(2 * D) + c);
        for (int k = 1; k < qs; ++k) {
          const float4 o = *reinterpret_cast<const float4 *>(dkv_part + ((size_t)k * T + t0) * (2 * D) + c);
          const float4 q = *reinterpret_cast<const float4 *>(dkv_part + ((size_t)k * T + t1) * (2 * D) + c);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          w.x += q.x; w.y += q.y; w.z += q.z; w.w += q.w;
        }
        *reinterpret_cast<float4 *>(out0 + D + c) = v;
        if (two) *reinterpret_cast<float4 *>(out1 + D + c) = w;
        am0 = max(am0, umax4(v));
        am1 = max(am1, umax4(w));
      }
    }
    if (row_scale) {
      const uint32_t s0 = pt_row_scale_bits(group_umax<64>(am0)), s1 = pt_row_scale_bits(group_umax<64>(am1));
      if (lane == 0) {
        atomicMin(row_scale + t0, s0);
        if (two) atomicMin(row_scale + t1, s1);
      }
      sb_min = two ? min(s0, s1) : s0;
    }
  }
  if (row_scale && row_min) {
    if (lane == 0) atomicMin(&sMin, sb_min);
    __syncthreads();
    if (threadIdx.x < 4) atomicMin(row_min + threadIdx.x, sMin);
  }
}

template <typename K>
static int set_lds(K kernel, int parts) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(parts * ATTN_LDS));
  if (e != hipSuccess) {
    g_pt_last_hip_error = e;
    return PTAMD_ERR_HIP;
  }
  return PTAMD_OK;
}
}  // namespace ptattn16

namespace ptgemm {
int persistent_grid(int reserved_cus);  // CUs of the current device (gemm.hip: a table filled once, no per-launch query)
}
namespace ptattn16 {
namespace {
// workgroup shape by the number of workgroups: 8 wavefronts (256 queries / keys); when those would cover at most half of
// the CUs, 8 wavefronts as 4 groups x 2 halves of the streamed dimension (128 per workgroup; W4 = 4 plain wavefronts for
// the dK/dV kernel, below); when even 128 per workgroup would, 4 wavefronts as 2 groups x 2 halves (64 per workgroup)
enum Shape { W8 = 0, W4 = 1, W4_HALVES = 2, W8_HALVES = 3 };
inline Shape launch_shape(int B, int L, int H) {
  const size_t cus = (size_t)ptgemm::persistent_grid(0), bh = (size_t)H * B;
  if ((size_t)((L + 255) / 256) * bh * 2 > cus) return W8;
  return (size_t)((L + 127) / 128) * bh * 2 > cus ? W8_HALVES : W4_HALVES;
}
// In between (as many 128-query workgroups as CUs, or up to twice as many CUs): 8 wavefronts as 4 query groups x 2 halves
// for the forward and dQ kernels - the same number of workgroups as with 4 wavefronts, half the tile loop, two
// wavefronts per SIMD; the dK/dV kernel keeps 4 wavefronts there (it has no registers left for a second staged tile
// pair at 8: 256 VGPRs already; built with 11 spilled registers it measured 0.5 % of a step).
inline Shape dkv_shape(Shape sh) { return sh == W8_HALVES ? W4 : sh; }

// the 2 x 4 forward shape instead of 2 x 2 where the key range has a tile for every quarter (PTAMD_ATTN_FWD_QUARTERS = 0 in the
// environment, read at every call: never - for A/B measurements)
inline bool fwd_quarters(int L) {
  if (const char *e = getenv("PTAMD_ATTN_FWD_QUARTERS")) return e[0] != '0' && L > 3 * TR;
  return L > 3 * TR;
}
template <int DK, int NW, int PARTS>
int launch_fwd(const float *qkv, const int64_t *seq, int B, int L, int H, float p, uint64_t seed, uint32_t sid, float *out,
               float *lse, uint32_t *keep_bits, hipStream_t st) {
  constexpr int QB = 32 * NW / PARTS;
  const dim3 grid((L + QB - 1) / QB, H, B);
  if (int rc = set_lds(attn_fwd_f16x2_kernel<DK, NW, PARTS>, PARTS)) return rc;  // idempotent, host-only: no state kept between calls
  hipLaunchKernelGGL((attn_fwd_f16x2_kernel<DK, NW, PARTS>), grid, dim3(64 * NW), PARTS * ATTN_LDS, st, qkv, seq, L, H, p, seed, sid,
                     out, lse, keep_bits);
  return pt_check_launch();
}
// 8 wavefronts as 2 query groups x 4 key quarters on unpadded tiles (round 6; head size 64): as many workgroups as 4 x (2 x 2)
// - 64 queries each - with HALF the tile loop per wavefront and TWO wavefronts per SIMD (one wavefront per SIMD runs a tile in
// 3.3 us, two in 1.5 each: the forward pass of the 4-protein share)
int launch_fwd_quarters(const float *qkv, const int64_t *seq, int B, int L, int H, float p, uint64_t seed, uint32_t sid, float *out,
                        float *lse, uint32_t *keep_bits, hipStream_t st) {
  constexpr size_t LDS = (size_t)4 * 2 * (2 * Tile2U::ELEMS) * sizeof(unsigned short);
  auto kern = attn_fwd_f16x2_kernel<64, 8, 4, Tile2U>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
  if (e != hipSuccess) {
    g_pt_last_hip_error = e;
    return PTAMD_ERR_HIP;
  }
  hipLaunchKernelGGL(kern, dim3((L + 63) / 64, H, B), dim3(512), LDS, st, qkv, seq, L, H, p, seed, sid, out, lse, keep_bits);
  return pt_check_launch();
}
template <int DK, int NW, int PARTS>
int launch_dq(const float *qkv, const int64_t *seq, const float *o_fwd, const float *d_o, const float *lse, float *delta,
              int B, int L, int H, float p, uint64_t seed, uint32_t sid, float *dqkv, uint32_t *row_scale, uint32_t *row_min,
              hipStream_t st) {
  constexpr int QB = 32 * NW / PARTS;
  const dim3 grid((L + QB - 1) / QB, H, B);
  if (int rc = set_lds(attn_bwd_dq_f16x2_kernel<DK, NW, PARTS>, PARTS)) return rc;
  hipLaunchKernelGGL((attn_bwd_dq_f16x2_kernel<DK, NW, PARTS>), grid, dim3(64 * NW), PARTS * ATTN_LDS, st, qkv, seq, o_fwd, d_o, lse,
                     delta, L, H, p, seed, sid, dqkv, row_scale, row_min);
  return pt_check_launch();
}
template <int DK, int NW, int PARTS>
int launch_dkv(const float *qkv, const int64_t *seq, const float *d_o, const float *lse, const float *delta, int B, int L, int H,
               float p, uint64_t seed, uint32_t sid, float *dqkv, uint32_t *row_scale, uint32_t *row_min, const uint32_t *keep_bits,
               hipStream_t st) {
  constexpr int QB = 32 * NW / PARTS;
  const dim3 grid((L + QB - 1) / QB, H, B);
  if (keep_bits && p > 0.f) {
    if (int rc = set_lds(attn_bwd_dkv_f16x2_kernel<DK, NW, PARTS, true>, PARTS)) return rc;
    hipLaunchKernelGGL((attn_bwd_dkv_f16x2_kernel<DK, NW, PARTS, true>), grid, dim3(64 * NW), PARTS * ATTN_LDS, st, qkv, seq, d_o, lse,
                       delta, L, H, p, seed, sid, dqkv, row_scale, row_min, keep_bits);
  } else {
    if (int rc = set_lds(attn_bwd_dkv_f16x2_kernel<DK, NW, PARTS, false>, PARTS)) return rc;
    hipLaunchKernelGGL((attn_bwd_dkv_f16x2_kernel<DK, NW, PARTS, false>), grid, dim3(64 * NW), PARTS * ATTN_LDS, st, qkv, seq, d_o, lse,
                       delta, L, H, p, seed, sid, dqkv, row_scale, row_min, nullptr);
  }
  return pt_check_launch();
}
template <int TPS>
int launch_fwd_kvp_t(const float *qkv, const char *kvp, const float *kv_inv, const int64_t *seq, int B, int L, int H, float p,
                     uint64_t seed, uint32_t sid, float *out, float *lse, uint32_t *keep_bits, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_fwd_kvp_f16x2_kernel<TPS>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)KvpGeo<TPS>::LDS);
  if (e != hipSuccess) {
    g_pt_last_hip_error = e;
    return PTAMD_ERR_HIP;
  }
  hipLaunchKernelGGL(attn_fwd_kvp_f16x2_kernel<TPS>, dim3((L + 255) / 256, H, B), dim3(512), KvpGeo<TPS>::LDS, st, qkv, kvp, kv_inv,
                     (B * L) / 32, seq, L, H, p, seed, sid, out, lse, keep_bits);
  return pt_check_launch();
}
// 64-key stages (one barrier per two tiles); PTAMD_ATTN_KVP_TPS = 1 in the environment (read at every call) selects 32-key
// stages for A/B measurements - same results either way
int launch_fwd_kvp(const float *qkv, const char *kvp, const float *kv_inv, const int64_t *seq, int B, int L, int H, float p,
                   uint64_t seed, uint32_t sid, float *out, float *lse, uint32_t *keep_bits, hipStream_t st) {
  const char *e = getenv("PTAMD_ATTN_KVP_TPS");
  if (e && e[0] == '1') return launch_fwd_kvp_t<1>(qkv, kvp, kv_inv, seq, B, L, H, p, seed, sid, out, lse, keep_bits, st);
  return launch_fwd_kvp_t<2>(qkv, kvp, kv_inv, seq, B, L, H, p, seed, sid, out, lse, keep_bits, st);
}
template <int DK>
int fwd_by_shape(Shape sh, const float *qkv, const int64_t *seq, int B, int L, int H, float p, uint64_t seed, uint32_t sid,
                 float *out, float *lse, uint32_t *keep_bits, hipStream_t st) {
  if (sh == W8) return launch_fwd<DK, 8, 1>(qkv, seq, B, L, H, p, seed, sid, out, lse, keep_bits, st);
  if (sh == W8_HALVES) return launch_fwd<DK, 8, 2>(qkv, seq, B, L, H, p, seed, sid, out, lse, keep_bits, st);
  if (DK == 64 && fwd_quarters(L)) return launch_fwd_quarters(qkv, seq, B, L, H, p, seed, sid, out, lse, keep_bits, st);
  return launch_fwd<DK, 4, 2>(qkv, seq, B, L, H, p, seed, sid, out, lse, keep_bits, st);
}
// the fused kernel: dk = 64 and enough (protein, head) pairs that one workgroup each fills more than half of the chip
// (PTAMD_ATTN_FUSED = 0 / 1 in the environment, read at every call: never / whenever dk = 64 - for tests, which run small
// batches, and for A/B measurements; the choice changes the summation order of dQ, nothing else)
// PTAMD_ATTN_FUSED in the environment (read at every call; for tests, which run small batches, and for A/B measurements):
// 0 = never (the two-kernel path), 1 = the unsplit sweep whatever the batch, 2 = the split sweep whatever the batch.  The
// choice changes the summation order of dQ (two-kernel path) / of dK and dV (split ranges), nothing else.
inline bool use_fused(int B, int L, int H, int dk) {   // the UNSPLIT one-sweep kernel: one workgroup per (protein, head)
  if (dk != 64) return false;
  if (const char *e = getenv("PTAMD_ATTN_FUSED")) return e[0] == '1';
  return (size_t)B * H * 2 > (size_t)ptgemm::persistent_grid(0);
}
// The split sweep (round 6): head size 64 and too few (protein, head) pairs for one workgroup each to fill the chip - the
// per-GPU share of a strongly scaled batch (4 / 8 / 16 proteins x 512: 92 / ~135 / ~205 us of dQ + dK/dV kernels per layer).
// split = 1: a workgroup per (pair, 256-key block); 2: the query tiles cut into `qs` ranges too, so that about one
// workgroup per CU comes out.
struct FusedSplit {
  int split, nkb, qs;
};
inline FusedSplit fused_split(int B, int L, int H, int dk) {
  FusedSplit f = {0, (L + FK - 1) / FK, 1};
  if (dk != 64) return f;
  if (const char *e = getenv("PTAMD_ATTN_FUSED")) {
    if (e[0] != '2') return f;
  } else if (use_fused(B, L, H, dk)) {
    return f;
  }
  const size_t cus = (size_t)ptgemm::persistent_grid(0), wg = (size_t)B * H * f.nkb;
  const int ntiles = (L + TR - 1) / TR;
  int qs = 1;
  while ((size_t)(2 * qs) * wg <= cus && 2 * qs <= ntiles) qs *= 2;
  const int per = (ntiles + qs - 1) / qs;
  f.qs = (ntiles + per - 1) / per;      // ranges that are not empty
  f.split = f.qs > 1 ? 2 : 1;
  return f;
}
inline size_t fused_split_floats(int B, int L, int H, const FusedSplit &f) {   // slabs behind delta in the workspace
  if (!f.split) return 0;
  const size_t T = (size_t)B * L, D = (size_t)H * 64;
  return (size_t)f.nkb * T * D + (f.split == 2 ? (size_t)f.qs * T * 2 * D : 0);
}
int launch_fused_split(const FusedSplit &f, const float *qkv, const int64_t *seq, const float *o_fwd, const float *d_o,
                       const float *lse, float *delta, float *slabs, int B, int L, int H, float p, uint64_t seed, uint32_t sid,
                       float *dqkv, uint32_t *row_scale, uint32_t *row_min, const uint32_t *keep_bits, const char *kvp,
                       const float *kv_inv, hipStream_t st) {
  const size_t items = (size_t)B * L * H * 16, T = (size_t)B * L;
  const int D = H * 64;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, o_fwd, d_o, B * L, L, H, 64, delta);
  const bool bits = keep_bits != nullptr && p > 0.f;
  float *dq_part = slabs, *dkv_part = slabs + (size_t)f.nkb * T * D;
  // (pre-split K / V only where the 256-query forward kernel reads them too: one workgroup per key block, no query ranges)
  auto kern = f.split == 2 ? (bits ? attn_bwd_fused_f16x2_kernel<true, false, 2> : attn_bwd_fused_f16x2_kernel<false, false, 2>)
              : kvp        ? (bits ? attn_bwd_fused_f16x2_kernel<true, true, 1> : attn_bwd_fused_f16x2_kernel<false, true, 1>)
                           : (bits ? attn_bwd_fused_f16x2_kernel<true, false, 1> : attn_bwd_fused_f16x2_kernel<false, false, 1>);
  if (kvp && f.split == 2) return PTAMD_ERR_BAD_SHAPE;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_LDS);
  if (e != hipSuccess) {
    g_pt_last_hip_error = e;
    return PTAMD_ERR_HIP;
  }
  hipLaunchKernelGGL(kern, dim3(f.nkb * f.qs, H, B), dim3(512), FUSED_LDS, st, qkv, seq, d_o, lse, delta, L, H, p, seed, sid, dqkv,
                     row_scale, row_min, keep_bits, kvp, kv_inv, (B * L) / 32, dq_part, dkv_part, f.qs);
  int rc = pt_check_launch();
  if (rc) return rc;
  const dim3 rgrid((unsigned)((T + 7) / 8));
  if (f.split == 2)
    hipLaunchKernelGGL(attn_bwd_split_reduce_kernel<true>, rgrid, dim3(256), 0, st, dq_part, f.nkb, dkv_part, f.qs, T, D, dqkv,
                       row_scale, row_min);
  else
    hipLaunchKernelGGL(attn_bwd_split_reduce_kernel<false>, rgrid, dim3(256), 0, st, dq_part, f.nkb, dkv_part, f.qs, T, D, dqkv,
                       row_scale, row_min);
  return pt_check_launch();
}
int launch_fused(const float *qkv, const int64_t *seq, const float *o_fwd, const float *d_o, const float *lse, float *delta,
                 int B, int L, int H, float p, uint64_t seed, uint32_t sid, float *dqkv, uint32_t *row_scale,
                 uint32_t *row_min, const uint32_t *keep_bits, const char *kvp, const float *kv_inv, hipStream_t st) {
  const size_t items = (size_t)B * L * H * 16;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, o_fwd, d_o, B * L, L, H, 64, delta);
  const bool bits = keep_bits != nullptr && p > 0.f;
  auto kern = kvp ? (bits ? attn_bwd_fused_f16x2_kernel<true, true> : attn_bwd_fused_f16x2_kernel<false, true>)
                  : (bits ? attn_bwd_fused_f16x2_kernel<true, false> : attn_bwd_fused_f16x2_kernel<false, false>);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_LDS);
  if (e != hipSuccess) {
    g_pt_last_hip_error = e;
    return PTAMD_ERR_HIP;
  }
  hipLaunchKernelGGL(kern, dim3(1, H, B), dim3(512), FUSED_LDS, st, qkv, seq, d_o, lse, delta, L, H, p, seed, sid, dqkv, row_scale,
                     row_min, keep_bits, kvp, kv_inv, (B * L) / 32, (float *)nullptr, (float *)nullptr, 1);
  return pt_check_launch();
}
template <int DK>
int bwd_by_shape(Shape sh, const float *qkv, const int64_t *seq, const float *o_fwd, const float *d_o, const float *lse,
                 float *delta, int B, int L, int H, float p, uint64_t seed, uint32_t sid, float *dqkv, uint32_t *row_scale,
                 uint32_t *row_min, const uint32_t *keep_bits, hipStream_t st) {
  int rc;
  if (sh == W8) rc = launch_dq<DK, 8, 1>(qkv, seq, o_fwd, d_o, lse, delta, B, L, H, p, seed, sid, dqkv, row_scale, row_min, st);
  else if (sh == W8_HALVES) rc = launch_dq<DK, 8, 2>(qkv, seq, o_fwd, d_o, lse, delta, B, L, H, p, seed, sid, dqkv, row_scale, row_min, st);
  else rc = launch_dq<DK, 4, 2>(qkv, seq, o_fwd, d_o, lse, delta, B, L, H, p, seed, sid, dqkv, row_scale, row_min, st);
  if (rc) return rc;
  const Shape kv = dkv_shape(sh);
  if (kv == W8) return launch_dkv<DK, 8, 1>(qkv, seq, d_o, lse, delta, B, L, H, p, seed, sid, dqkv, row_scale, row_min, keep_bits, st);
  if (kv == W4) return launch_dkv<DK, 4, 1>(qkv, seq, d_o, lse, delta, B, L, H, p, seed, sid, dqkv, row_scale, row_min, keep_bits, st);
  return launch_dkv<DK, 4, 2>(qkv, seq, d_o, lse, delta, B, L, H, p, seed, sid, dqkv, row_scale, row_min, keep_bits, st);
}
}  // namespace
}  // namespace ptattn16

// pre-split K / V (kv_format.h) are read by the 256-query forward kernel and the one-sweep backward kernel: head size 64, whole
// 32-token tiles per protein, and the batch shapes at which exactly those two kernels run
bool pt_attention_f16x2_reads_kv_planes(int B, int L, int H, int dk) {
  using namespace ptattn16;
  // (round 6: also the split sweep with one workgroup per key block - 16 proteins x 8 heads x 512 - whose forward pass is the
  // 256-query kernel as well)
  if (!(B > 0 && L > 0 && H > 0 && dk == 64 && (L & 31) == 0 && launch_shape(B, L, H) == W8)) return false;
  return use_fused(B, L, H, dk) || fused_split(B, L, H, dk).split == 1;
}

int pt_attention_fwd_f16x2(const float *qkv, const int64_t *seq, int B, int L, int H, int dk, float p, uint64_t seed,
                           uint32_t sid, float *out, float *lse, uint32_t *keep_bits, const void *kv_planes, const float *kv_inv,
                           hipStream_t st) {
  using namespace ptattn16;
  if (kv_planes) {
    if (!kv_inv || !pt_attention_f16x2_reads_kv_planes(B, L, H, dk)) return PTAMD_ERR_BAD_SHAPE;
    return launch_fwd_kvp(qkv, static_cast<const char *>(kv_planes), kv_inv, seq, B, L, H, p, seed, sid, out, lse, keep_bits, st);
  }
  const Shape sh = launch_shape(B, L, H);
  return dk == 64 ? fwd_by_shape<64>(sh, qkv, seq, B, L, H, p, seed, sid, out, lse, keep_bits, st)
                  : fwd_by_shape<32>(sh, qkv, seq, B, L, H, p, seed, sid, out, lse, keep_bits, st);
}

// (the one-sweep kernel and, on the two-kernel path, the dK / dV kernel - both keep keys in lanes; the dQ kernel draws them)
bool pt_attention_bwd_f16x2_reads_keep_bits(int B, int L, int H, int dk) { return B > 0 && L > 0 && H > 0 && (dk == 64 || dk == 32); }

// floats of workspace the f16x2 backward pass of this shape wants BEHIND delta (the slabs of the split sweep; 0 otherwise)
size_t pt_attention_bwd_f16x2_slab_floats(int B, int L, int H, int dk) {
  using namespace ptattn16;
  if (B <= 0 || L <= 0 || H <= 0) return 0;
  return fused_split_floats(B, L, H, fused_split(B, L, H, dk));
}

int pt_attention_bwd_f16x2(const float *qkv, const int64_t *seq, const float *o_fwd, const float *d_o, const float *lse,
                           float *delta, int B, int L, int H, int dk, float p, uint64_t seed, uint32_t sid, float *dqkv,
                           uint32_t *row_scale, uint32_t *row_min, const uint32_t *keep_bits, const void *kv_planes,
                           const float *kv_inv, float *slabs, size_t slab_floats, hipStream_t st) {
  using namespace ptattn16;
  if (kv_planes && (!kv_inv || !pt_attention_f16x2_reads_kv_planes(B, L, H, dk))) return PTAMD_ERR_BAD_SHAPE;
  // (the forward kernel's decisions are read by the fused kernel and by the dK / dV kernel of the two-kernel path)
  if (use_fused(B, L, H, dk))
    return launch_fused(qkv, seq, o_fwd, d_o, lse, delta, B, L, H, p, seed, sid, dqkv, row_scale, row_min, keep_bits,
                        static_cast<const char *>(kv_planes), kv_inv, st);
  const FusedSplit fs = fused_split(B, L, H, dk);
  if (fs.split) {
    if (!slabs || slab_floats < fused_split_floats(B, L, H, fs)) return PTAMD_ERR_WORKSPACE;
    return launch_fused_split(fs, qkv, seq, o_fwd, d_o, lse, delta, slabs, B, L, H, p, seed, sid, dqkv, row_scale, row_min, keep_bits,
                              static_cast<const char *>(kv_planes), kv_inv, st);
  }
  const Shape sh = launch_shape(B, L, H);
  return dk == 64 ? bwd_by_shape<64>(sh, qkv, seq, o_fwd, d_o, lse, delta, B, L, H, p, seed, sid, dqkv, row_scale, row_min, keep_bits, st)
                  : bwd_by_shape<32>(sh, qkv, seq, o_fwd, d_o, lse, delta, B, L, H, p, seed, sid, dqkv, row_scale, row_min, keep_bits, st);
}
