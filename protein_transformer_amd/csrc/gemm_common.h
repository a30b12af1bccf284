// Pieces shared by the two MFMA GEMM kernels of libptamd (gemm.hip: exact f32 MFMA; gemm_split_kernel.h: f32 operands
// split into three bf16 terms on the bf16 MFMA pipe): launch parameters, the global->register stage loader, the work
// decomposition of the persistent workgroups and the fused epilogue.
#pragma once
#include "common.h"

namespace ptgemm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, NT = 256;  // output tile per workgroup, threads per workgroup
// BK (K advance per stage) is a template parameter: 32 (2 workgroups / CU, 73.7 KB LDS each) or 16 (3 per CU).
// [row][k] image of a K-contiguous operand: BK k + 4 pad floats per row (36 or 20): conflict-free ds_read_b128
// [k][row] image of a row-contiguous operand: 128 + 4 floats per k: straight 16-byte copies, ds_read_b32
constexpr int LD_C = 132;

struct GemmParams {
  int M, N, K;
  const float *A;
  int lda;
  const float *B;
  int ldb;
  float *C;
  int ldc;
  const float *bias;
  const float *residual;
  int ldr;
  int flags;
  float dropout_p;
  uint64_t seed;
  uint32_t stream_id;
  int k_per_split;  // multiple of BK
  int splits;
  size_t slab;      // M*N when split-K writes partial slabs, else 0
  int vec_epilogue;  // N, ldc, ldr multiples of 4 and 16-byte aligned C / residual: float4 epilogue through LDS
  float *colsum;     // k-major A only: colsum[m] (+)= sum_k A[k][m]; with split-K a [splits * share][M] slab, reduced later
  int colsum_share;  // N tiles sharing the column-sum work of one (M tile, split): power of two <= min(tiles_n, 16)
  float gate_scale;  // PTAMD_EPI_GATE
  const uint32_t *scale_a, *scale_b;  // f16x2 arithmetic only: power-of-two scale (bits) per row of A / column of B
  int scale_a_stride, scale_b_stride; // 1: one scale per operand row; 0: ONE scale for every row (a caller-provided bound; the
                                      // array then holds four copies, because row-contiguous operands load four rows' scales at once)
  int reserved_cus;  // CUs the persistent grid leaves free (room for a concurrent collective kernel); 0 = none
  // PTAMD_EPI_GATE from a 1-bit gate instead of the [M, N] fp32 activation (include/ptamd.h, ptamd_gate_mask_bytes): entry
  // ((cb * mask_rb + rb) * 16 + r) = the 64 lane decisions of accumulator register r of the 32 x 32 block (rb, cb) - bit l =
  // element (row 32 rb + (r & 3) + 8 (r >> 2) + 4 (l >> 5), column 32 cb + (l & 31)) passes.  Written by the epilogue of the
  // product whose ReLU + dropout it gates (gate_mask_out: the ballot of "result > 0"), read as scalar loads + v_cndmask.
  const uint64_t *gate_mask;
  uint64_t *gate_mask_out;
  int mask_rb, mask_cb;  // 32-row / 32-column blocks of the masked matrix
};
// The 16 gate masks of the 32 x 32 block (rb, cb) - 128 bytes - arrive by ONE coalesced vector load: lane l holds dword
// l & 31 (the compiler will not use scalar loads here: the kernel stores to global memory in front of them and the pointer
// sits in a by-value struct, so it cannot prove the words unclobbered).  Register r's 64 lane decisions are then two
// v_readlane into a scalar pair and the select is a v_cndmask on that pair.  Blocks outside the matrix read block 0:
// nothing of them is ever stored.
__device__ __forceinline__ uint32_t load_gate_masks(const GemmParams &p, int rb, int cb, int lane) {
  const int rbu = __builtin_amdgcn_readfirstlane(rb), cbu = __builtin_amdgcn_readfirstlane(cb);
  const bool in = rbu < p.mask_rb && cbu < p.mask_cb;
  const uint32_t *src = reinterpret_cast<const uint32_t *>(p.gate_mask + ((size_t)(in ? cbu : 0) * p.mask_rb + (in ? rbu : 0)) * 16);
  return src[lane & 31];
}
__device__ __forceinline__ bool gate_keep(uint32_t masks, int r) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)masks, 2 * r), hi = (uint32_t)__builtin_amdgcn_readlane((int)masks, 2 * r + 1);
  return __builtin_amdgcn_inverse_ballot_w64(((uint64_t)hi << 32) | lo);
}

// ---- staging: each thread carries 4 float4 per operand per stage; global -> registers -> LDS, no transposition:
//      the LDS image keeps the operand's own contiguity and the MFMA k-assignment adapts instead
//      (MFMA step (m, j) of a stage uses k = 8m + 4*(lane>>5) + j for BOTH operands).
template <bool KMAJOR, int BK>
__device__ __forceinline__ void load_stage(const float *__restrict__ src, int ld, int rows, int r0, int kend, int k0,
                                           int tid, float4 (&v)[BK / 8]) {
  constexpr int LPR = BK / 4;  // lanes per row of a K-contiguous operand
#pragma unroll
  for (int i = 0; i < BK / 8; ++i) {
    if (!KMAJOR) {  // src[row][k]: BK/4 lanes cover the BK k of one row
      const int row = r0 + tid / LPR + (NT / LPR) * i, k = k0 + 4 * (tid % LPR);
      v[i] = (row < rows && k < kend) ? *reinterpret_cast<const float4 *>(src + (size_t)row * ld + k)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {  // src[k][row]: 32 lanes cover 128 consecutive rows of one k (512 B)
      const int k = k0 + (tid >> 5) + 8 * i, row = r0 + 4 * (tid & 31);
      v[i] = (k < kend && row < rows) ? *reinterpret_cast<const float4 *>(src + (size_t)k * ld + row)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}
__device__ __forceinline__ float epilogue_value(float v, int row, int col, const GemmParams &p, uint32_t thr,
                                                float keep_scale) {
  if (p.bias) v += p.bias[col];
  if (p.flags & PTAMD_EPI_RELU) v = fmaxf(v, 0.f);
  if (p.dropout_p > 0.f) v = drop_keep(p.seed, p.stream_id, row, col, p.N, thr >> 16) ? v * keep_scale : 0.f;
  if (p.residual) {
    const float r = p.residual[(size_t)row * p.ldr + col];
    v = (p.flags & PTAMD_EPI_GATE) ? (r > 0.f ? v * p.gate_scale : 0.f) : v + r;
  }
  if (p.flags & PTAMD_EPI_TANH) v = tanhf(v);
  return v;
}


// ---- epilogue of one 128 x 128 tile held as 2 x 2 MFMA accumulators per wavefront (wm, wn = wavefront row / column).
// C/D map of the 32x32 MFMA (same for the f32 and the bf16 instruction): col = lane & 31,
// row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
//
// Vector path: bias / ReLU / dropout are applied in the MFMA layout (one column per lane, four consecutive rows per
// generator call), then each wavefront transposes its tile through `scratch` (its own 2048 floats of LDS, 32 rows x 64
// columns at a time) so that residual / accumulate operands are READ and results are WRITTEN as float4 rows:
// 16 16-byte stores per lane instead of 64 4-byte ones.
// (row0, col0) = origin of this wavefront's (32 TI) x 64 block of C.
// VEC = false: same transposition, but the four elements of a lane are read / written one by one (any N / ldc).
// EPI selects how much of the epilogue is compiled in (instruction-cache footprint: the generator alone is most of the
// code): EPI_PLAIN stores the accumulators as they are, EPI_NODROP has everything but dropout, EPI_FULL everything.
constexpr int EPI_PLAIN = 0, EPI_NODROP = 1, EPI_FULL = 2;
// MASK_GATE: the 1-bit gate (GemmParams::gate_mask) is compiled in - the f16x2 no-dropout instantiation of the staging
// kernel only (the bf16x3 one sits at 256 registers: with the mask code it spilled 200 of them and ran five times slower)
template <int TI, bool VEC = true, int EPI = EPI_FULL, bool MASK_GATE = false>
__device__ __forceinline__ void tile_epilogue_vec(const GemmParams &p, const f32x16 (&acc)[TI][2], float *C, int ldc,
                                                  bool partial, int row0, int col0, int lane, uint32_t thr,
                                                  float keep_scale, float *scratch) {
  const int l31 = lane & 31, lh = lane >> 5;
  // Dropout keep decisions of the whole block, one bit per accumulator element (bit j * 16 + g * 4 + e of keep[i]),
  // drawn in a ROLLED loop: one copy of the generator in the instruction stream instead of 8 TI.
  uint32_t keep[TI];
#pragma unroll
  for (int i = 0; i < TI; ++i) keep[i] = 0xffffffffu;
  if (EPI == EPI_FULL && !partial && p.dropout_p > 0.f) {
#pragma unroll
    for (int i = 0; i < TI; ++i) keep[i] = 0u;
#pragma unroll 4
    for (int idx = 0; idx < TI * 4; ++idx) {  // one call = the lane's 8 rows of two register groups (common.h)
      const int i = idx >> 2, j = (idx >> 1) & 1, gp = idx & 1;
      const int row = row0 + i * 32 + 16 * gp + 4 * lh, col = col0 + j * 32 + l31;
      const uint4 rnd = pt_rand4(p.seed, drop_call_index(row, col, p.N), p.stream_id);
      const uint32_t t16 = thr >> 16;
      uint32_t bits = 0;
#pragma unroll
      for (int f = 0; f < 8; ++f)   // field f = (g & 1) * 4 + e -> accumulator register (2 gp + (f >> 2)) * 4 + (f & 3)
        bits |= (drop_field_value(rnd, f) >= t16 ? 1u : 0u) << (j * 16 + (2 * gp + (f >> 2)) * 4 + (f & 3));
#pragma unroll
      for (int ii = 0; ii < TI; ++ii) keep[ii] |= ii == i ? bits : 0u;
    }
  }
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    // residual / gate / accumulate operands of this 32-row block are requested BEFORE the transposition, so that their
    // latency runs under the LDS traffic instead of in front of every store
    f32x4 pre4[8];  // the residual / gate operand if there is one, else the old C of an accumulating product
    const bool want_res = VEC && EPI != EPI_PLAIN && !partial && p.residual != nullptr;
    const bool want_old = VEC && EPI != EPI_PLAIN && !partial && (p.flags & PTAMD_EPI_ACCUM) && !want_res;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int f = lane + 64 * t, rr = f >> 4, c4 = (f & 15) * 4;
      const int row = min(row0 + i * 32 + rr, p.M - 1), col = min(col0 + c4, p.N - 4);
      const float *src = want_res ? p.residual + (size_t)row * p.ldr + col : C + (size_t)row * ldc + col;
      pre4[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (want_res || want_old) pre4[t] = *reinterpret_cast<const f32x4 *>(src);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = col0 + j * 32 + l31;
      const float bias = (EPI != EPI_PLAIN && !partial && p.bias && col < p.N) ? p.bias[col] : 0.f;
      // (uniform) the 1-bit gate of this 32 x 32 block, applied in the accumulator layout: one coalesced load of its 16
      // masks, two v_readlane + one select per element
      const bool gated = VEC && MASK_GATE && !partial && p.gate_mask != nullptr;
      uint32_t gm = 0;
      if (gated) gm = load_gate_masks(p, (row0 >> 5) + i, (col0 >> 5) + j, lane);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int rowq = row0 + i * 32 + 8 * g + 4 * lh;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[i][j][g * 4 + e];
          if (EPI != EPI_PLAIN && !partial) {
            v += bias;
            if (p.flags & PTAMD_EPI_RELU) v = fmaxf(v, 0.f);
            if (EPI == EPI_FULL && p.dropout_p > 0.f) v = (keep[i] >> (j * 16 + g * 4 + e)) & 1u ? v * keep_scale : 0.f;
            if (gated) v = gate_keep(gm, g * 4 + e) ? v * p.gate_scale : 0.f;
          }
          scratch[(8 * g + 4 * lh + e) * 64 + j * 32 + l31] = v;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int f = lane + 64 * t, rr = f >> 4, c4 = (f & 15) * 4;
      const int row = row0 + i * 32 + rr, col = col0 + c4;
      float4 v = *reinterpret_cast<const float4 *>(scratch + rr * 64 + c4);
      if (VEC) {
#ifdef PT_ABLATE_NOSTORE   // ablation build: everything but the global store of the tile (the condition is never true)
        if (row < p.M && col < p.N && p.M < 0) {
#else
        if (row < p.M && col < p.N) {
#endif
          if (EPI != EPI_PLAIN && !partial) {
            if (p.residual) {
              const f32x4 r4 = pre4[t];
              if (p.flags & PTAMD_EPI_GATE) {
                v.x = r4.x > 0.f ? v.x * p.gate_scale : 0.f; v.y = r4.y > 0.f ? v.y * p.gate_scale : 0.f;
                v.z = r4.z > 0.f ? v.z * p.gate_scale : 0.f; v.w = r4.w > 0.f ? v.w * p.gate_scale : 0.f;
              } else {
                v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
              }
            }
            if (p.flags & PTAMD_EPI_TANH) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
            if (p.flags & PTAMD_EPI_ACCUM) {
              const f32x4 o4 = want_old ? pre4[t] : *reinterpret_cast<const f32x4 *>(C + (size_t)row * ldc + col);
              v.x += o4.x; v.y += o4.y; v.z += o4.z; v.w += o4.w;
            }
          }
          *reinterpret_cast<float4 *>(C + (size_t)row * ldc + col) = v;
        }
      } else {
        const float ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (row < p.M && col + e < p.N) {
            float x = ve[e];
            if (EPI != EPI_PLAIN && !partial) {
              if (p.residual) {
                const float r = p.residual[(size_t)row * p.ldr + col + e];
                x = (p.flags & PTAMD_EPI_GATE) ? (r > 0.f ? x * p.gate_scale : 0.f) : x + r;
              }
              if (p.flags & PTAMD_EPI_TANH) x = tanhf(x);
              if (p.flags & PTAMD_EPI_ACCUM) x += C[(size_t)row * ldc + col + e];
            }
            C[(size_t)row * ldc + col + e] = x;
          }
        }
      }
    }
  }
}

// Work decomposition of a persistent launch: logical id = ((z * tiles_m + tm) * tiles_n + tn), i.e. neighbours share
// the same A panel / K chunk.  Workgroup b (which the dispatcher places on XCD b % 8) owns the CONTIGUOUS logical
// range [w_begin, w_end) with slot = (b % 8) * (G / 8) + b / 8: consecutive items of one workgroup and the
// workgroups of one XCD all walk neighbouring tiles, so an A panel is fetched through one L2 and re-read by the
// same few CUs instead of being requested by every N tile at once.
struct WorkRange {
  int tiles_m, tiles_n, begin, end, bm, bn;
  __device__ __forceinline__ WorkRange(const GemmParams &p, int tile_m = BM, int tile_n = BN) : bm(tile_m), bn(tile_n) {
    tiles_n = (p.N + bn - 1) / bn;
    tiles_m = (p.M + bm - 1) / bm;
    const int nwork = tiles_m * tiles_n * p.splits;
    const int G = gridDim.x, base = nwork / G, rem = nwork - base * G;
    const int slot = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    begin = slot * base + min(slot, rem);
    end = begin + base + (slot < rem ? 1 : 0);
  }
  __device__ __forceinline__ void decode(int logical, int &bm0, int &bn0, int &z) const {
    const int ntile = tiles_m * tiles_n;
    z = logical / ntile;
    const int tile = logical - z * ntile;
    bm0 = (tile / tiles_n) * bm;
    bn0 = (tile % tiles_n) * bn;
  }
};

// ---- a GROUP of independent products in one persistent launch (ptamd_gemm_group: the weight-gradient products of a layer).
// The work items of the members are concatenated (member 0's first); a workgroup owns a contiguous range of the whole list
// exactly as in WorkRange, so a launch of four members with 768 items is three items per workgroup whatever the members' sizes.
constexpr int MAX_GROUP = 4;
struct GemmGroup {
  GemmParams p[MAX_GROUP];
  int first[MAX_GROUP + 1];  // member j owns the logical ids [first[j], first[j + 1])
  int n;
};
struct GroupRange {
  int begin, end, bm, bn;
  __device__ __forceinline__ GroupRange(const GemmGroup &g, int tile_m, int tile_n) : bm(tile_m), bn(tile_n) {
    const int nwork = g.first[g.n];
    const int G = gridDim.x, base = nwork / G, rem = nwork - base * G;
    const int slot = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    begin = slot * base + min(slot, rem);
    end = begin + base + (slot < rem ? 1 : 0);
  }
  // logical id -> member, tile origin, K split (inside a member: as WorkRange::decode)
  __device__ __forceinline__ void decode(const GemmGroup &g, int logical, int &member, int &bm0, int &bn0, int &z) const {
    member = 0;
#pragma unroll
    for (int j = 1; j < MAX_GROUP; ++j) member += (j < g.n && logical >= g.first[j]) ? 1 : 0;
    const GemmParams &p = g.p[member];
    const int tiles_n = (p.N + bn - 1) / bn, ntile = ((p.M + bm - 1) / bm) * tiles_n, local = logical - g.first[member];
    z = local / ntile;
    const int tile = local - z * ntile;
    bm0 = (tile / tiles_n) * bm;
    bn0 = (tile % tiles_n) * bn;
  }
};

int persistent_grid(int reserved_cus);  // CUs of the current device minus the reserve, at least 1 (gemm.hip)
// a group of k-major x k-major products in f16x2 arithmetic writing split-K slabs (gemm_f16x2.hip), and the fixed-order
// reduction of all their slabs in one launch (gemm.hip)
int launch_group_f16x2(const GemmGroup &g, hipStream_t st);
struct ReduceMember {
  GemmParams p;  // C = the user's matrix again, slab = M N
  const float *slabs, *cs_slabs;
  float *colsum;
  int splits, first_block;
};
struct ReduceGroup {
  ReduceMember m[MAX_GROUP];
  int n, blocks;
};
int launch_splitk_reduce_group(const ReduceGroup &g, hipStream_t st);
// launchers of the two kernels: a_kmajor / b_kmajor select the instantiation
int launch_f32(const GemmParams &p, bool a_kmajor, bool b_kmajor, int splits, hipStream_t st);
int launch_split(const GemmParams &p, bool a_kmajor, bool b_kmajor, int splits, int products, hipStream_t st);
int launch_split_f16x2(const GemmParams &p, bool a_kmajor, bool b_kmajor, int splits, hipStream_t st);
// fixed-order reduction of split-K slabs into p.C with the epilogue of p (gemm.hip)
int launch_splitk_reduce(const GemmParams &p, const float *slabs, int splits, const float *cs_slabs, float *colsum, hipStream_t st);
// f16x2 arithmetic: fills scale_a[M] / scale_b[N] (device, uint32 bits of powers of two) from the operands of p
// (a NULL scale pointer skips that operand: its scales were provided by the caller)
int launch_row_scales(const GemmParams &p, bool a_kmajor, bool b_kmajor, uint32_t *scale_a, uint32_t *scale_b, hipStream_t st);

}  // namespace ptgemm
