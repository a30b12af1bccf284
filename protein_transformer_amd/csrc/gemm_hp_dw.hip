// ptamd_gemm_hp_dw: the weight-gradient product of a torch.nn.Linear from TOKEN-MAJOR pre-split operands (hp_format.h):
//     dW[M, N] (+)= sum_t dY[t, m] X[t, n],     dbias[m] (+)= sum_t dY[t, m]
// (the backward of Attention.py:38-41,49,69 / Sublayers.py:28-34 of the reference: what autograd computes for
// `F.linear(x, W, b)` - dW = dy^T x, db = sum dy).  Both operands are the SAME buffers the forward / dX products read as
// their K-contiguous A operand: planes of [T tokens, features] in 32-token x 16-feature blocks.  Here the contraction runs
// over the tokens, i.e. over the ROWS of the blocks, so
//   * a stage = one block row (32 tokens) of both operands: 32 KiB of dY (256 features) + 16 KiB of X (128 features),
//     contiguous in memory, filled by LDS-DMA exactly like in gemm_hp.hip (no VGPRs, no VALU, no ds_write);
//   * fragments are read with `ds_read_b64_tr_b16` (LDS transpose read: 4 tokens x 16 features per 16-lane group -> every
//     lane gets 4 consecutive tokens of its feature).  The two 16-feature blocks of a 32-feature MFMA tile would sit 2 KiB
//     apart and hit the same banks, so the DMA places every odd block 128 bytes further (a block PAIR occupies 4352 bytes):
//     both halves of a 32-lane service group then cover all 64 banks - conflict-free;
//   * the per-row (= per-token) scales of the two operands sit INSIDE the contraction and cannot be taken out of the
//     accumulators.  A small pre-kernel turns them into one power of two per token, fac[t] = c / (sY[t] sX[t]) <= 1 with
//     c = min_t sY[t] sX[t], in f16; the X fragments are multiplied by it (v_pk_mul_f16, exact up to f16 underflow -
//     a token whose dY row is 2^k below the largest loses k of its low bits, which is what ONE scale for the whole
//     operand would have cost it) and the accumulators are divided by c at the end.  The error model is that of the
//     uniform-scale f16x2 weight-gradient products of ptamd_gemm (include/ptamd.h);
//   * the bias gradient is a v_dot2_f32_f16 per fragment register against gfac[t] = min_t sY / sY[t] in the wavefronts that
//     hold the dY fragments of the first N tile.
// Structure otherwise as gemm_hp.hip: one persistent 512-thread workgroup per CU, 256 x 128 output tile, 8 wavefronts of
// 64 x 64, two stage buffers, DMA pieces of stage q + 1 issued between the MFMA groups of stage q, split-K over the tokens
// with fixed-order slab reduction (gemm.hip), XCD-aware contiguous work ranges.  The accumulators are stored straight
// from the MFMA layout (full 128-byte lines per store instruction): a tile is written once per 16-64 stages here.
#include <stdlib.h>

#include "gemm_common.h"
#include "hp_format.h"
#include "split_bf16.h"

namespace pthpdw {
namespace {

using ptgemm::GemmParams;
using ptgemm::f32x16;
using ptsplit::f16x8;
using ptsplit::lds_s16x4;
using ptsplit::s16x4;
using ptsplit::s16x8;
using namespace pthp;

typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));

constexpr int TM = 256, TN = 128, NW = 8, THREADS = 512, BKT = 32;   // output tile, wavefronts, tokens per stage
constexpr int PAIR = 4352, ODD = 2176;                               // LDS bytes of a block pair / offset of its odd block
constexpr int A_BLOCKS = TM / 16, B_BLOCKS = TN / 16;                // 16-feature blocks of a stage
constexpr int A_BYTES = A_BLOCKS / 2 * PAIR, STAGE_BYTES = (A_BLOCKS + B_BLOCKS) / 2 * PAIR;
constexpr int PIECES = (A_BLOCKS + B_BLOCKS) * 2, PER_WAVE = PIECES / NW;   // 1-KiB DMA pieces of a stage
constexpr size_t LDS = (size_t)2 * STAGE_BYTES;
static_assert(ODD % 256 == 128 && PAIR % 256 == 0, "odd blocks must fall on the other half of the bank row");
static_assert(LDS <= 160 * 1024, "LDS budget of a CU");

struct DwParams {
  GemmParams g;                 // M, N, splits, k_per_split (tokens), C, ldc, slab, flags (ACCUM), reserved_cus
  const char *y, *x;            // planes of dY [T, M] and X [T, N]
  int ykb16, xkb16;             // 16-feature blocks per block row of each operand
  int y_kb_last, x_kb_last;     // last valid feature block (tiles beyond are clamped, never stored)
  int Tp;                       // tokens rounded up to 32
  const _Float16 *fac, *gfac;   // [Tp] per-token factors (pre-kernel below)
  const float *consts;          // [0] [1]: the two halves of 1 / c, [2]: 1 / min_t sY
  float *colsum;                // NULL, dbias [M] (splits == 1: accumulated in place) or its [splits][M] slabs
  int ablate;                   // PTAMD_DW_ABLATE (measurement only): 1 no DMA after the prologue, 2 no fragment reads, 4 all stages read block row 0
};

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
__device__ __forceinline__ void dma16(const char *g, char *l) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

struct Item {
  int bm0, bn0, z, kbeg, kend;
};
struct Cursor {
  int w, k0;
  Item it;
};

__device__ __forceinline__ f16x8 tr_frag(const char *q) {   // tokens +0..3 at q, +4..7 128 bytes further
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)q);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(q + 128));
  const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(f16x8, both);
}
__device__ __forceinline__ f16x8 scale_frag(f16x8 v, f16x8 f) { return v * f; }   // 4 x v_pk_mul_f16
__device__ __forceinline__ float dot_frag(f16x8 v, f16x8 f, float acc) {            // 4 x v_dot2_f32_f16
  acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(v, v, 0, 1), __builtin_shufflevector(f, f, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(v, v, 2, 3), __builtin_shufflevector(f, f, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(v, v, 4, 5), __builtin_shufflevector(f, f, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(v, v, 6, 7), __builtin_shufflevector(f, f, 6, 7), acc, false);
  return acc;
}

template <bool COLSUM>
__global__ __launch_bounds__(THREADS, 2) void gemm_hp_dw_kernel(const DwParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  // the wavefront index as a SCALAR: everything derived from it (which DMA piece, which operand, LDS destinations) is then
  // computed on the scalar unit and the operand base pointer is a scalar select - left as a vector value the compiler
  // re-loaded the selected pointer from the kernel-argument segment inside every DMA slot and waited vmcnt(0) for it,
  // draining the LDS-DMA queue in the middle of the stage
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const ptgemm::WorkRange work(p.g, TM, TN);
  if (work.begin >= work.end) return;
  auto item_at = [&](int logical) __attribute__((always_inline)) {
    Item it;
    work.decode(logical, it.bm0, it.bn0, it.z);
    it.kbeg = it.z * p.g.k_per_split;
    it.kend = min(p.Tp, it.kbeg + p.g.k_per_split);
    return it;
  };
  auto advance = [&](Cursor &c) __attribute__((always_inline)) {
    if (c.k0 + BKT < c.it.kend) {
      c.k0 += BKT;
      return true;
    }
    if (c.w + 1 < work.end) {
      c.it = item_at(++c.w);
      c.k0 = c.it.kbeg;
      return true;
    }
    return false;
  };
  // DMA piece i of this wavefront: piece q = wave + 8 i = (block j, plane) of the stage, 1 KiB contiguous in memory
  const int lane16 = lane * 16;
  auto issue_piece = [&](const Cursor &c, int buf, int i) __attribute__((always_inline)) {
    const int q = wave + NW * i;
    const bool is_b = q >= 2 * A_BLOCKS;
    const int jq = is_b ? q - 2 * A_BLOCKS : q, j = jq >> 1, plane = jq & 1;
    const int kb = is_b ? min((c.it.bn0 >> 4) + j, p.x_kb_last) : min((c.it.bm0 >> 4) + j, p.y_kb_last);
    const char *g = (is_b ? p.x : p.y) + block_offset((p.ablate & 4) ? 0 : c.k0 >> 5, kb, plane, is_b ? p.xkb16 : p.ykb16) + lane16;
    dma16(g, smem + buf * STAGE_BYTES + (is_b ? A_BYTES : 0) + (j >> 1) * PAIR + (j & 1) * ODD + plane * 1024);
  };

  f32x16 acc[2][2];
  float cs[2] = {0.f, 0.f};
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  zero_acc();
  const bool partial = p.g.slab != 0;
  // this lane's corner of every transpose read: tokens 8 kh + rr (+ 4), features 16 g + 4 c .. + 3 of a block pair
  const int q16 = lane & 15, kh = lane >> 5, rr = q16 >> 2, c4 = q16 & 3;
  const int lane_off = ((lane >> 4) & 1) * ODD + (2 * (8 * kh + rr) + ((c4 >> 1) ^ kh)) * 16 + (c4 & 1) * 8;

  Cursor ld = {work.begin, 0, item_at(work.begin)};
  ld.k0 = ld.it.kbeg;
  Cursor cc = ld;
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) issue_piece(ld, 0, i);
  bool more_loads = advance(ld);
  // per-token factors: 8 tokens of each of the two 16-token steps of a stage (this lane's k half).  They are fetched one
  // stage ahead, right behind the barrier that opens the stage before, so that their L2 latency runs under that stage
  // (fetched where they are used they sat in front of the `vmcnt(0)` of every stage: 2.1 instead of 1.x us per stage).
  f16x8 fac[2], gf[2], fac_n[2], gf_n[2];
  auto load_factors = [&](int k0, f16x8 (&f)[2], f16x8 (&g)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f[ks] = *reinterpret_cast<const f16x8 *>(p.fac + k0 + 16 * ks + 8 * kh);
      g[ks] = COLSUM ? *reinterpret_cast<const f16x8 *>(p.gfac + k0 + 16 * ks + 8 * kh) : f[ks];
    }
  };
  load_factors(cc.k0, fac, gf);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) { fac_n[ks] = fac[ks]; gf_n[ks] = gf[ks]; }
  int buf = 0;
  for (;;) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's pieces of the stage have landed ...
    __syncthreads();                                   // ... and everybody's; everybody is done with the other buffer
    if (more_loads) load_factors(ld.k0, fac_n, gf_n);  // `ld` is on the next stage
    const char *sa = smem + buf * STAGE_BYTES + (2 * wm) * PAIR + lane_off;
    const char *sb = smem + buf * STAGE_BYTES + A_BYTES + (2 * wn) * PAIR + lane_off;
    const bool cs_on = COLSUM && wn == 0 && cc.it.bn0 == 0;
    int piece = 0;
    // The two wavefronts of a SIMD (w and w + 4) issue their DMA pieces in DIFFERENT halves of the stage - wavefronts
    // 0-3 between the MFMA groups of the first 16-token step, 4-7 in the second - so that one of them is issuing MFMAs
    // while the other sits in its DMA issue (an LDS-DMA instruction holds its wavefront until the texture addresser
    // takes the 1 KiB; with all eight wavefronts in the same slot the matrix pipe of every SIMD idles meanwhile).
    auto dma_slot = [&](int ks, int n) __attribute__((always_inline)) {
      __builtin_amdgcn_sched_barrier(0);
      if (more_loads && !(p.ablate & 1) && ((p.ablate & 8) || (wave >> 2) == ks)) {
#pragma unroll
        for (int u = 0; u < n; ++u)
          if (piece + u < PER_WAVE) issue_piece(ld, buf ^ 1, piece + u);
        piece += n;
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // ALL fragments of the stage first, then the MFMAs with the DMA pieces of the next stage between them: a transpose
    // read that follows an LDS-DMA issue in program order gets an `s_waitcnt vmcnt(0)` from the compiler (it cannot tell
    // that the DMA writes the other buffer), which would drain the queue in the middle of the stage
    f16x8 fa[2][2][2], fb[2][2][2];  // [step][tile][plane]
    if (p.ablate & 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 2; ++i) { fa[ks][i][t] = fac[ks]; fb[ks][i][t] = gf[ks]; }
    } else
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[ks][i][t] = tr_frag(sa + i * PAIR + t * 1024 + ks * 512);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[ks][j][t] = tr_frag(sb + j * PAIR + t * 1024 + ks * 512);
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[ks][j][t] = scale_frag(fb[ks][j][t], fac[ks]);
      if (cs_on) {  // uniform per wavefront
#pragma unroll
        for (int i = 0; i < 2; ++i) cs[i] = dot_frag(fa[ks][i][0], gf[ks], dot_frag(fa[ks][i][1], gf[ks], cs[i]));
      }
      // smallest products first
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks][i][1], fb[ks][j][0], acc[i][j], 0, 0, 0);
      dma_slot(ks, (p.ablate & 8) ? 1 : 2);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks][i][0], fb[ks][j][1], acc[i][j], 0, 0, 0);
      dma_slot(ks, (p.ablate & 8) ? 1 : 2);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks][i][0], fb[ks][j][0], acc[i][j], 0, 0, 0);
      dma_slot(ks, (p.ablate & 8) ? 1 : 2);
    }
    if (more_loads) more_loads = advance(ld);
    if (cc.k0 + BKT >= cc.it.kend) {  // that was the item's last stage (uniform): acc / c, straight from the MFMA layout
      const float i1 = p.consts[0], i2 = p.consts[1];
      float *C = partial ? p.g.C + (size_t)cc.it.z * p.g.slab : p.g.C;
      const int ldc = partial ? p.g.N : p.g.ldc;
      const bool accum = !partial && (p.g.flags & PTAMD_EPI_ACCUM);
      const int l31 = lane & 31;
#pragma unroll 1
      for (int ij = 0; ij < 4; ++ij) {   // rolled: one copy of the store sequence in the instruction stream
        const int i = ij >> 1, j = ij & 1;
        const f32x16 a = ij == 0 ? acc[0][0] : ij == 1 ? acc[0][1] : ij == 2 ? acc[1][0] : acc[1][1];
        const int col = cc.it.bn0 + wn * 64 + j * 32 + l31;
        const int row0 = cc.it.bm0 + wm * 64 + i * 32 + 4 * kh;
        float *dst = C + (size_t)row0 * ldc + col;
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          old[r] = (accum && row0 + dr < p.g.M && col < p.g.N) ? dst[(size_t)dr * ldc] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          if (row0 + dr < p.g.M && col < p.g.N) dst[(size_t)dr * ldc] = a[r] * i1 * i2 + old[r];
        }
      }
      zero_acc();
      if (COLSUM && wn == 0 && cc.it.bn0 == 0) {
        const float ig = p.consts[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float v = (cs[i] + __shfl_xor(cs[i], 32, 64)) * ig;
          const int row = cc.it.bm0 + wm * 64 + i * 32 + l31;
          if (kh == 0 && row < p.g.M) {
            if (partial) p.colsum[(size_t)cc.it.z * p.g.M + row] = v;
            else p.colsum[row] += v;
          }
          cs[i] = 0.f;
        }
      }
    }
    if (!advance(cc)) break;
    buf ^= 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { fac[ks] = fac_n[ks]; gf[ks] = gf_n[ks]; }
  }
}

// fac[t] = 2^(emin - eY[t] - eX[t]) with emin = min_t (eY + eX), gfac[t] = 2^(eYmin - eY[t]) as f16 (0 below 2^-24), and
// consts = {2^-(emin / 2), 2^-(emin - emin / 2), 2^-eYmin}: one block, the T scales are read twice (128 KiB at T = 16384)
__device__ __forceinline__ unsigned short pow2_f16(int d) {   // 2^d, d <= 0
  return d >= -14 ? (unsigned short)((d + 15) << 10) : d >= -24 ? (unsigned short)(1u << (d + 24)) : (unsigned short)0;
}
__device__ __forceinline__ float pow2_f32(int e) {            // 2^e for any e (saturating to 0 / the largest power)
  e = max(-149, min(127, e));
  return e >= -126 ? __uint_as_float((uint32_t)(e + 127) << 23) : __uint_as_float(1u << (e + 149));
}
constexpr int FAC_BLOCKS = 16;
__global__ __launch_bounds__(1024) void hp_dw_factors_kernel(const uint32_t *__restrict__ sy, const uint32_t *__restrict__ sx,
                                                             int T, int Tp, unsigned short *__restrict__ fac,
                                                             unsigned short *__restrict__ gfac, float *__restrict__ consts) {
  // every block finds the two minima over ALL tokens itself (128 KiB of L2 reads at T = 16384, 16 independent loads per
  // thread in flight) and writes its own slice of the factors: no second launch, no atomics
  __shared__ int red[2][16];
  int emin = 1 << 20, ymin = 1 << 20;
  for (int t0 = 0; t0 < T; t0 += 16 * 1024) {
    uint32_t vy[16], vx[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int t = min(t0 + u * 1024 + (int)threadIdx.x, T - 1);
      vy[u] = sy[t];
      vx[u] = sx[t];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int ey = (int)(vy[u] >> 23) - 127, ex = (int)(vx[u] >> 23) - 127;
      emin = min(emin, ey + ex);
      ymin = min(ymin, ey);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    emin = min(emin, __shfl_xor(emin, o, 64));
    ymin = min(ymin, __shfl_xor(ymin, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = emin;
    red[1][threadIdx.x >> 6] = ymin;
  }
  __syncthreads();
  emin = red[0][0];
  ymin = red[1][0];
#pragma unroll
  for (int w = 1; w < 16; ++w) {
    emin = min(emin, red[0][w]);
    ymin = min(ymin, red[1][w]);
  }
  for (int t = blockIdx.x * 1024 + threadIdx.x; t < Tp; t += gridDim.x * 1024) {
    unsigned short f = 0, g = 0;
    if (t < T) {
      const int ey = (int)(sy[t] >> 23) - 127, ex = (int)(sx[t] >> 23) - 127;
      f = pow2_f16(emin - ey - ex);
      g = pow2_f16(ymin - ey);
    }
    fac[t] = f;
    gfac[t] = g;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int h = emin / 2;
    consts[0] = pow2_f32(-h);
    consts[1] = pow2_f32(-(emin - h));
    consts[2] = pow2_f32(-ymin);
    consts[3] = 0.f;
  }
}

size_t slab_floats(int M, int N, int splits) { return splits > 1 ? (size_t)splits * M * N + (size_t)splits * M : 0; }
size_t align16(size_t b) { return (b + 15) & ~(size_t)15; }

}  // namespace
}  // namespace pthpdw

using namespace pthpdw;

extern "C" {

size_t ptamd_gemm_hp_dw_workspace_bytes(int M, int N, int T, int split_k) {
  if (M <= 0 || N <= 0 || T <= 0) return 0;
  const int Tp = round_up(T, 32);
  return align16(slab_floats(M, N, split_k > 1 ? split_k : 1) * sizeof(float)) + align16((size_t)2 * Tp * sizeof(unsigned short)) + 16;
}

int ptamd_gemm_hp_dw(const ptamd_gemm_hp_dw_args *a, void *stream) {
  if (!a || a->M <= 0 || a->N <= 0 || a->T <= 0) return PTAMD_ERR_BAD_SHAPE;
  if (!a->Y || !a->X || !a->Y_scale || !a->X_scale || !a->C || !a->workspace) return PTAMD_ERR_BAD_SHAPE;
  if (!pt_aligned16(a->Y) || !pt_aligned16(a->X) || !pt_aligned16(a->workspace)) return PTAMD_ERR_ALIGN;
  const int Tp = round_up(a->T, 32), stages = Tp / BKT;
  int splits = a->split_k > 1 ? a->split_k : 1;
  if (splits > stages) splits = stages;
  const int kps = ((stages + splits - 1) / splits) * BKT;
  splits = (Tp + kps - 1) / kps;
  if (a->workspace_bytes < ptamd_gemm_hp_dw_workspace_bytes(a->M, a->N, a->T, splits)) return PTAMD_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char *ws = static_cast<char *>(a->workspace);
  float *slabs = reinterpret_cast<float *>(ws);
  const size_t slab_bytes = align16(slab_floats(a->M, a->N, splits) * sizeof(float));
  unsigned short *fac = reinterpret_cast<unsigned short *>(ws + slab_bytes);
  unsigned short *gfac = fac + Tp;
  float *consts = reinterpret_cast<float *>(ws + slab_bytes + align16((size_t)2 * Tp * sizeof(unsigned short)));
  hipLaunchKernelGGL(hp_dw_factors_kernel, dim3(min(FAC_BLOCKS, (Tp + 1023) / 1024)), dim3(1024), 0, st, reinterpret_cast<const uint32_t *>(a->Y_scale),
                     reinterpret_cast<const uint32_t *>(a->X_scale), a->T, Tp, fac, gfac, consts);
  DwParams p;
  GemmParams &g = p.g;
  g.M = a->M; g.N = a->N; g.K = Tp;
  g.A = nullptr; g.lda = 0; g.B = nullptr; g.ldb = 0; g.C = a->C; g.ldc = a->ldc;
  g.bias = nullptr; g.residual = nullptr; g.ldr = 0; g.flags = a->accumulate ? PTAMD_EPI_ACCUM : 0;
  g.dropout_p = 0.f; g.seed = 0; g.stream_id = 0; g.gate_scale = 0.f;
  g.reserved_cus = a->reserved_cus;
  g.colsum = nullptr; g.colsum_share = 1; g.scale_a = g.scale_b = nullptr; g.scale_a_stride = g.scale_b_stride = 1;
  g.k_per_split = kps; g.splits = splits;
  g.vec_epilogue = 1;
  g.slab = 0;
  p.colsum = a->colsum;
  if (splits > 1) {
    g.slab = (size_t)a->M * a->N;
    g.C = slabs;
    if (a->colsum) p.colsum = slabs + (size_t)splits * g.slab;
  }
  p.y = static_cast<const char *>(a->Y);
  p.x = static_cast<const char *>(a->X);
  p.ykb16 = kb16(a->M);
  p.xkb16 = kb16(a->N);
  p.y_kb_last = p.ykb16 - 1;
  p.x_kb_last = p.xkb16 - 1;
  p.Tp = Tp;
  p.fac = reinterpret_cast<const _Float16 *>(fac);
  p.gfac = reinterpret_cast<const _Float16 *>(gfac);
  p.consts = consts;
  p.ablate = getenv("PTAMD_DW_ABLATE") ? atoi(getenv("PTAMD_DW_ABLATE")) : 0;
  const int work = ((a->M + TM - 1) / TM) * ((a->N + TN - 1) / TN) * splits;
  const int slots = ptgemm::persistent_grid(a->reserved_cus);
  const int grid = work < slots ? work : slots;
  if (a->colsum) {
    auto kern = gemm_hp_dw_kernel<true>;
    PT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), LDS, st, p);
  } else {
    auto kern = gemm_hp_dw_kernel<false>;
    PT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), LDS, st, p);
  }
  const int rc = pt_check_launch();
  if (rc || splits == 1) return rc;
  // fixed-order sum of the slabs (+ C when accumulating) and of the bias-gradient slabs
  g.C = a->C;
  const bool vec_ok = !(a->N & 3) && !(a->ldc & 3) && pt_aligned16(a->C);
  g.vec_epilogue = vec_ok;
  return ptgemm::launch_splitk_reduce(g, slabs, splits, a->colsum ? slabs + (size_t)splits * g.slab : nullptr, a->colsum, st);
}

}  // extern "C"
