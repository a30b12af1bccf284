// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 products and accumulation,
// 64 FLOP/clk/SIMD = 157 TF/s chip peak) with fused bias / ReLU / dropout / residual / tanh epilogues.
//
// Replaces every torch.nn.Linear of the reference encoder and their autograd backward GEMMs:
//   wq/wk/wv/wo     /root/reference/protein_transformer/models/transformer/Attention.py:38-41,49,69
//   pwff.layer1/2   .../models/transformer/Sublayers.py:28-34
//   output_projection + tanh  .../models/encoder_only.py:18,39-41
//   residual + dropout of SublayerConnection  .../models/transformer/Sublayers.py:16-17
//
// Tiling (MI355X-first, wave64): workgroup = 128 x 128 outputs, 4 wavefronts in a 2 x 2 grid, each wavefront
// owns 64 x 64 = 2 x 2 MFMA tiles of 32 x 32 (64 accumulator VGPRs).  K advances 32 per stage through a
// double-buffered LDS image that keeps each operand's OWN contiguity (16-byte global loads -> 16-byte LDS
// writes, nothing is transposed): K-contiguous operands sit as [row][32+4] and are read back with one
// conflict-free ds_read_b128 per four MFMA steps, row-contiguous operands sit as [k][128+4] and are read with
// ds_read_b32.  The f32 MFMA needs ONE operand dword per lane per 64 cycles, so the k index a lane half feeds to
// MFMA step (m, j) is free to be k = 8m + 4*(lane>>5) + j for both operands.  Workgroups are persistent
// (2 per CU) and walk (tile, K-split) items; the next item's first stage is prefetched under the current
// item's last stage and epilogue.  Item ids are remapped so that the tiles of one 128-row panel of A run on the
// same XCD (shared L2).
#include <stdlib.h>

#include "gemm_common.h"

namespace ptgemm {
namespace {

template <bool KMAJOR, int BK>
__device__ __forceinline__ void store_stage(float *__restrict__ s, int tid, const float4 (&v)[BK / 8]) {
  constexpr int LPR = BK / 4, LD_R = BK + 4;
#pragma unroll
  for (int i = 0; i < BK / 8; ++i) {
    if (!KMAJOR) {
      const int row = tid / LPR + (NT / LPR) * i, k = 4 * (tid % LPR);
      *reinterpret_cast<float4 *>(s + row * LD_R + k) = v[i];
    } else {
      const int k = (tid >> 5) + 8 * i, row = 4 * (tid & 31);
      *reinterpret_cast<float4 *>(s + k * LD_C + row) = v[i];
    }
  }
}
// fragment of rows [r0 + l31] for the 4 MFMA steps of k-group m
template <bool KMAJOR, int BK>
__device__ __forceinline__ void read_frag(const float *__restrict__ s, int r, int lh, int m, float (&f)[4]) {
  constexpr int LD_R = BK + 4;
  if (!KMAJOR) {
    const float4 v = *reinterpret_cast<const float4 *>(s + r * LD_R + 8 * m + 4 * lh);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
    const float *q = s + (8 * m + 4 * lh) * LD_C + r;
    f[0] = q[0]; f[1] = q[LD_C]; f[2] = q[2 * LD_C]; f[3] = q[3 * LD_C];
  }
}

// Persistent workgroups: a launch has min(#work items, 2 per CU) workgroups, each walking a contiguous range of
// work items (output tile x K split).  The first K stage of the NEXT item is prefetched into
// registers/LDS before the epilogue of the current one, so the epilogue's stores overlap the next loads and
// the matrix pipe does not wait for a cold prologue per tile.
template <bool A_KMAJOR, bool B_KMAJOR, int BK>
__global__ __launch_bounds__(NT, 2) void gemm_f32_mfma_kernel(const GemmParams p) {
  constexpr int LD_R = BK + 4, NG = BK / 8;
  constexpr int SA = A_KMAJOR ? BK * LD_C : BM * LD_R, SB = B_KMAJOR ? BK * LD_C : BN * LD_R;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *const sA0 = smem;  // two stages of A, then two stages of B
  float *const sB0 = smem + 2 * SA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
  const WorkRange work(p);
  const int w_begin = work.begin, w_end = work.end;
  auto decode = [&](int logical, int &bm0, int &bn0, int &z) { work.decode(logical, bm0, bn0, z); };

  float4 ra[NG], rb[NG];
  int w = w_begin, bm0, bn0, z;
  if (w >= w_end) return;
  decode(w, bm0, bn0, z);
  int kbeg = z * p.k_per_split, kend = min(p.K, kbeg + p.k_per_split);
  load_stage<A_KMAJOR, BK>(p.A, p.lda, p.M, bm0, kend, kbeg, tid, ra);
  load_stage<B_KMAJOR, BK>(p.B, p.ldb, p.N, bn0, kend, kbeg, tid, rb);
  store_stage<A_KMAJOR, BK>(sA0, tid, ra);
  store_stage<B_KMAJOR, BK>(sB0, tid, rb);
  __syncthreads();
  int cur = 0;

  const bool partial = p.slab != 0;
  const uint32_t thr = dropout_threshold(p.dropout_p);
  const float keep_scale = 1.f / (1.f - p.dropout_p);

  while (true) {
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Bias gradient = column sums of the k-major A operand.  With split-K slabs the N tiles of one (M tile, split)
    // share the work: N tile tn takes every cs_share-th k row starting at tn (cs_share = power of two <= 16), and
    // the slab reduction adds the shares; without slabs the first N tile does it alone and accumulates in place.
    const int cs_share = partial ? p.colsum_share : 1, cs_first = (bn0 / BN) & (cs_share - 1);
    const bool do_colsum = A_KMAJOR && BK == 32 && p.colsum != nullptr && (partial ? bn0 / BN < cs_share : bn0 == 0);
    float csum = 0.f;
    const int wn_next = w + 1;
    int nbm0 = 0, nbn0 = 0, nz = 0, nkbeg = 0, nkend = 0;
    const bool has_next = wn_next < w_end;
    if (has_next) {
      decode(wn_next, nbm0, nbn0, nz);
      nkbeg = nz * p.k_per_split;
      nkend = min(p.K, nkbeg + p.k_per_split);
    }

    // Software pipeline over k-groups of 8 (16 MFMAs per wavefront): the fragments of group g+1 are read from
    // LDS while the MFMAs of group g execute; the next stage is written to the other LDS buffer after group 1
    // (its global loads were issued at the top of the stage); the ONE barrier of the stage sits before the
    // MFMAs of the last group, whose operands are already in registers, so those 16 MFMAs (1024 cycles) cover
    // the barrier skew and the LDS latency of the next stage's first fragments.
    const int ar = wm * 64 + l31, br = wn * 64 + l31;
    float a0[4], a1[4], b0[4], b1[4];
    {
      const float *sa = sA0 + cur * SA, *sb = sB0 + cur * SB;
      read_frag<A_KMAJOR, BK>(sa, ar, lh, 0, a0);
      read_frag<A_KMAJOR, BK>(sa, ar + 32, lh, 0, a1);
      read_frag<B_KMAJOR, BK>(sb, br, lh, 0, b0);
      read_frag<B_KMAJOR, BK>(sb, br + 32, lh, 0, b1);
    }
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      const bool more = k0 + BK < kend;
      const bool fetch = more || has_next;
      if (more) {  // next stage of this item
        load_stage<A_KMAJOR, BK>(p.A, p.lda, p.M, bm0, kend, k0 + BK, tid, ra);
        load_stage<B_KMAJOR, BK>(p.B, p.ldb, p.N, bn0, kend, k0 + BK, tid, rb);
      } else if (has_next) {  // first stage of the next item: flies under this item's last stage + epilogue
        load_stage<A_KMAJOR, BK>(p.A, p.lda, p.M, nbm0, nkend, nkbeg, tid, ra);
        load_stage<B_KMAJOR, BK>(p.B, p.ldb, p.N, nbn0, nkend, nkbeg, tid, rb);
      }
      const float *sa = sA0 + cur * SA, *sb = sB0 + cur * SB;
      const float *na = sA0 + (cur ^ 1) * SA, *nb = sB0 + (cur ^ 1) * SB;
      if (A_KMAJOR && do_colsum) {  // sum over (this workgroup's share of) the stage's k of A[k][m]
        const float *q = sa + ((tid >> 7) * (BK / 2) + cs_first) * LD_C + (tid & 127);
        for (int kk = 0; kk < BK / 2; kk += cs_share) csum += q[kk * LD_C];
      }
#pragma unroll
      for (int m = 0; m < NG; ++m) {
        float c0[4], c1[4], d0[4], d1[4];
        if (m < NG - 1) {
          read_frag<A_KMAJOR, BK>(sa, ar, lh, m + 1, c0);
          read_frag<A_KMAJOR, BK>(sa, ar + 32, lh, m + 1, c1);
          read_frag<B_KMAJOR, BK>(sb, br, lh, m + 1, d0);
          read_frag<B_KMAJOR, BK>(sb, br + 32, lh, m + 1, d1);
        } else {
          __syncthreads();  // next buffer fully written by every wave; nobody reads `cur` any more
          if (fetch) {
            read_frag<A_KMAJOR, BK>(na, ar, lh, 0, c0);
            read_frag<A_KMAJOR, BK>(na, ar + 32, lh, 0, c1);
            read_frag<B_KMAJOR, BK>(nb, br, lh, 0, d0);
            read_frag<B_KMAJOR, BK>(nb, br + 32, lh, 0, d1);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
        }
        if (m == (NG >= 4 ? 1 : 0) && fetch) {
          store_stage<A_KMAJOR, BK>(sA0 + (cur ^ 1) * SA, tid, ra);
          store_stage<B_KMAJOR, BK>(sB0 + (cur ^ 1) * SB, tid, rb);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a0[j] = c0[j]; a1[j] = c1[j]; b0[j] = d0[j]; b1[j] = d1[j];
        }
      }
      cur ^= 1;
    }

    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
    float *C = p.C + (partial ? (size_t)z * p.slab : 0);
    const int ldc = partial ? p.N : p.ldc;
    {
      // Bias / ReLU / dropout are applied in the MFMA layout (one column per lane), then each wavefront transposes its
      // tile through the LDS stage buffer that the last K stage just released (32 rows x 64 columns at a time) so
      // that residual / accumulate operands are READ and results are WRITTEN as float4 rows: 16 16-byte stores per
      // lane instead of 64 4-byte ones (element by element when N or a leading dimension is not a multiple of 4).
      float *scratch = (wave < 2 ? sA0 + (cur ^ 1) * SA : sB0 + (cur ^ 1) * SB) + (wave & 1) * 2048;
      if (p.vec_epilogue)
        tile_epilogue_vec<2, true>(p, acc, C, ldc, partial, bm0 + wm * 64, bn0 + wn * 64, lane, thr, keep_scale, scratch);
      else
        tile_epilogue_vec<2, false>(p, acc, C, ldc, partial, bm0 + wm * 64, bn0 + wn * 64, lane, thr, keep_scale, scratch);
      if (has_next) __syncthreads();  // the next item's first stage store reuses this LDS buffer
    }
    if (A_KMAJOR && do_colsum) {  // block-uniform
      float *spare = sA0 + (cur ^ 1) * SA + 4096;  // 128 floats the wave scratch regions do not use
      __syncthreads();
      if (tid >= 128) spare[tid - 128] = csum;
      __syncthreads();
      if (tid < 128 && bm0 + tid < p.M) {
        const float tot = csum + spare[tid];
        if (partial) p.colsum[((size_t)z * cs_share + cs_first) * p.M + bm0 + tid] = tot;
        else p.colsum[bm0 + tid] += tot;
      }
      __syncthreads();
    }
    if (!has_next) break;
    w = wn_next; bm0 = nbm0; bn0 = nbn0; z = nz; kbeg = nkbeg; kend = nkend;
  }
}

// split-K, plain case (no bias / activation / dropout / residual; optional accumulate) with 16-byte accesses: one
// thread sums one float4 of one output row over the slabs in a fixed order.  The column-sum slabs (fused bias gradient
// of a dW product) are reduced by the first wavefronts of the grid.
__device__ __forceinline__ void splitk_reduce_plain(const GemmParams &p, const float *__restrict__ slabs, int splits,
                                                    const float *__restrict__ colsum_slabs, float *__restrict__ colsum_out,
                                                    size_t i) {
  const int n4 = p.N >> 2;
  const size_t total = (size_t)p.M * n4;
  if (colsum_out) {  // M rows, one lane per (row, 16th of the slabs): 16 lanes per row
    const int nslab = splits * p.colsum_share;
    const size_t g = i >> 4;
    if (g < (size_t)p.M) {
      const int chunk = (int)(i & 15);
      float v = 0.f;
      for (int s = chunk; s < nslab; s += 16) v += colsum_slabs[(size_t)s * p.M + g];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
      if (chunk == 0) colsum_out[g] += v;
    }
  }
  if (i >= total) return;
  const size_t row = i / n4;
  const int c4 = (int)(i - row * n4) * 4;
  const float4 *src = reinterpret_cast<const float4 *>(slabs) + i;
  const size_t stride4 = p.slab >> 2;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int s = 0;
  for (; s + 4 <= splits; s += 4) {  // four independent loads in flight, summed in slab order
    const float4 a = src[(size_t)s * stride4], b = src[(size_t)(s + 1) * stride4], c = src[(size_t)(s + 2) * stride4],
                 d = src[(size_t)(s + 3) * stride4];
    acc.x = (((acc.x + a.x) + b.x) + c.x) + d.x;
    acc.y = (((acc.y + a.y) + b.y) + c.y) + d.y;
    acc.z = (((acc.z + a.z) + b.z) + c.z) + d.z;
    acc.w = (((acc.w + a.w) + b.w) + c.w) + d.w;
  }
  for (; s < splits; ++s) {
    const float4 a = src[(size_t)s * stride4];
    acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
  }
  float4 *dst = reinterpret_cast<float4 *>(p.C + row * p.ldc + c4);
  if (p.flags & PTAMD_EPI_ACCUM) {
    const float4 o = *dst;
    acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
  }
  *dst = acc;
}
__global__ __launch_bounds__(256) void gemm_splitk_reduce_plain_kernel(const GemmParams p, const float *__restrict__ slabs,
                                                                     int splits, const float *__restrict__ colsum_slabs,
                                                                     float *__restrict__ colsum_out) {
  splitk_reduce_plain(p, slabs, splits, colsum_slabs, colsum_out, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}
// the same for the members of a group launch: blocks [first_block[j], first_block[j + 1]) work on member j
__global__ __launch_bounds__(256) void gemm_splitk_reduce_group_kernel(const ReduceGroup g) {
  int j = 0;
#pragma unroll
  for (int k = 1; k < MAX_GROUP; ++k) j += (k < g.n && (int)blockIdx.x >= g.m[k].first_block) ? 1 : 0;
  const ReduceMember &m = g.m[j];
  splitk_reduce_plain(m.p, m.slabs, m.splits, m.cs_slabs, m.colsum,
                      (size_t)((int)blockIdx.x - m.first_block) * blockDim.x + threadIdx.x);
}

// split-K: sum the slabs in a fixed order, then the same epilogue
__global__ void gemm_splitk_reduce_kernel(const GemmParams p, const float *__restrict__ slabs, int splits,
                                          const float *__restrict__ colsum_slabs, float *__restrict__ colsum_out) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int rowq = blockIdx.y * 4;
  if (colsum_out && blockIdx.x == 0 && threadIdx.x < 64) {  // bias gradient of rows rowq..rowq+3: one wavefront,
    const int e = threadIdx.x & 3, chunk = threadIdx.x >> 2;   // 16 lanes per row, each summing every 16th slab
    const int row = rowq + e, nslab = splits * p.colsum_share;
    float v = 0.f;
    if (row < p.M)
      for (int s = chunk; s < nslab; s += 16) v += colsum_slabs[(size_t)s * p.M + row];
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    if (chunk == 0 && row < p.M) colsum_out[row] += v;
  }
  if (col >= p.N) return;
  const uint32_t thr = dropout_threshold(p.dropout_p);
  const float keep_scale = 1.f / (1.f - p.dropout_p);
  for (int e = 0; e < 4; ++e) {
    const int row = rowq + e;
    if (row >= p.M) return;
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += slabs[(size_t)s * p.slab + (size_t)row * p.N + col];
    v = epilogue_value(v, row, col, p, thr, keep_scale);
    if (p.flags & PTAMD_EPI_ACCUM) v += p.C[(size_t)row * p.ldc + col];
    p.C[(size_t)row * p.ldc + col] = v;
  }
}

}  // namespace

namespace {
// CU counts of the visible devices: immutable facts about the hardware, queried once (thread-safe static initialisation) so
// that the ~100 GEMM launches of a step do not each pay a hipDeviceGetAttribute - not state: nothing a call does changes it
constexpr int MAX_DEVICES = 64;
struct CuTable {
  int cus[MAX_DEVICES];
  CuTable() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    for (int d = 0; d < MAX_DEVICES; ++d) {
      int c = 0;
      if (d >= n || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || c <= 0) c = 256;
      cus[d] = c;
    }
  }
};
}  // namespace

int persistent_grid(int reserved_cus) {  // CUs of the current device minus the reserve, at least 1
  static const CuTable table;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
  int cus = table.cus[dev];
  if (reserved_cus > 0) cus -= reserved_cus;
  return cus > 0 ? cus : 1;
}

namespace {
template <bool AK, bool BKM>
int launch(const GemmParams &p, int splits, hipStream_t st) {
  constexpr int BK = 32, LD_R = BK + 4;
  constexpr int SA = AK ? BK * LD_C : BM * LD_R, SB = BKM ? BK * LD_C : BN * LD_R;
  const size_t lds = (size_t)2 * (SA + SB) * sizeof(float);
  const int work = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * splits;
  auto kern = gemm_f32_mfma_kernel<AK, BKM, BK>;
  // idempotent and host-only: set on every launch so that the library keeps no state between calls
  PT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int slots = persistent_grid(p.reserved_cus) * 2;
  const int grid = work < slots ? work : slots;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, p);
  return pt_check_launch();
}
}  // namespace

int launch_f32(const GemmParams &p, bool a_kmajor, bool b_kmajor, int splits, hipStream_t st) {
  if (!a_kmajor && !b_kmajor) return launch<false, false>(p, splits, st);
  if (!a_kmajor && b_kmajor) return launch<false, true>(p, splits, st);
  if (a_kmajor && !b_kmajor) return launch<true, false>(p, splits, st);
  return launch<true, true>(p, splits, st);
}

// fixed-order sum of the split-K slabs (+ the epilogue; + the column-sum slabs of a fused bias gradient) into p.C
int launch_splitk_reduce(const GemmParams &p, const float *slabs, int splits, const float *cs_slabs, float *colsum,
                         hipStream_t st) {
  const bool plain = !p.bias && !p.residual && !(p.flags & (PTAMD_EPI_RELU | PTAMD_EPI_TANH)) && p.dropout_p == 0.f;
  if (plain && !(p.N & 3) && !(p.ldc & 3) && pt_aligned16(p.C) && pt_aligned16(slabs)) {
    const size_t work = (size_t)p.M * (p.N >> 2), cs_work = colsum ? (size_t)p.M * 16 : 0;
    const size_t threads = work > cs_work ? work : cs_work;
    hipLaunchKernelGGL(gemm_splitk_reduce_plain_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, p, slabs,
                       splits, cs_slabs, colsum);
  } else {
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((p.N + 255) / 256, (p.M + 3) / 4), dim3(256), 0, st, p, slabs,
                       splits, cs_slabs, colsum);
  }
  return pt_check_launch();
}

int launch_splitk_reduce_group(const ReduceGroup &g, hipStream_t st) {
  hipLaunchKernelGGL(gemm_splitk_reduce_group_kernel, dim3(g.blocks), dim3(256), 0, st, g);
  return pt_check_launch();
}

}  // namespace ptgemm

using namespace ptgemm;

extern "C" {

namespace {
size_t slab_bytes(int M, int N, int split_k) {  // C slabs + column-sum slabs of a split-K product, rounded to 16 bytes
  if (split_k <= 1) return 0;
  return (((size_t)split_k * M * N + (size_t)split_k * 16 * M) * sizeof(float) + 15) & ~(size_t)15;
}
int round4(int n) { return (n + 3) & ~3; }
}  // namespace

size_t ptamd_gemm_workspace_bytes(int M, int N, int split_k) {
  if (M <= 0 || N <= 0) return 0;
  return slab_bytes(M, N, split_k) + (size_t)(round4(M) + round4(N)) * sizeof(uint32_t);  // + row scales (f16x2 arithmetic)
}

namespace {
bool valid_arith(int mode) {
  return mode == PTAMD_GEMM_F32 || mode == PTAMD_GEMM_BF16X3 || mode == PTAMD_GEMM_BF16X3_FULL || mode == PTAMD_GEMM_F16X2 ||
         mode == PTAMD_GEMM_AUTO;
}
// the arithmetic one call runs in: the call's own `arith` field (there is no process-wide mode), AUTO resolved by operand
// layout, and the exact-f32 kernel for what the split kernels do not take (they address an operand with 32-bit byte
// offsets and start a K tail 16 k before its end)
int resolve_mode(const ptamd_gemm_args *a) {
  int mode = a->arith;
  // AUTO: f16x2 where the pass over the operands is cheap next to the product (K-contiguous A: activations x weights),
  // bf16x3 for the long token reductions of the weight gradients (both operands are read whole by that pass)
  if (mode == PTAMD_GEMM_AUTO) mode = a->a_kmajor ? PTAMD_GEMM_BF16X3 : PTAMD_GEMM_F16X2;
  const size_t a_bytes = (size_t)(a->a_kmajor ? a->K : a->M) * a->lda * sizeof(float);
  const size_t b_bytes = (size_t)(a->b_kmajor ? a->K : a->N) * a->ldb * sizeof(float);
  if (mode == PTAMD_GEMM_F16X2 && a->K < 32) mode = PTAMD_GEMM_BF16X3;   // the f16x2 kernel stages 32 k at a time
  if (a->K < 16 || a_bytes >= ((size_t)1 << 32) || b_bytes >= ((size_t)1 << 32)) mode = PTAMD_GEMM_F32;
  return mode;
}
}  // namespace

int ptamd_gemm_products(const ptamd_gemm_args *a) {
  if (!a || a->M <= 0 || a->N <= 0 || a->K <= 0 || !valid_arith(a->arith)) return PTAMD_ERR_BAD_SHAPE;
  const int mode = resolve_mode(a);
  return mode == PTAMD_GEMM_F32 ? 1 : mode == PTAMD_GEMM_F16X2 ? 3 : mode == PTAMD_GEMM_BF16X3_FULL ? 9 : 6;
}

namespace {
// arguments -> launch parameters: checks, K splits, slabs in the workspace; the f16x2 row scales are left to the caller
int build_params(const ptamd_gemm_args *a, GemmParams &p, int &splits, int &mode, float *&user_c) {
  if (!a || a->M <= 0 || a->N <= 0 || a->K <= 0 || !valid_arith(a->arith)) return PTAMD_ERR_BAD_SHAPE;
  if ((a->lda & 3) || (a->ldb & 3)) return PTAMD_ERR_BAD_SHAPE;
  if ((!a->a_kmajor || !a->b_kmajor) && (a->K & 3)) return PTAMD_ERR_BAD_SHAPE;  // K-contiguous rows are read 16 B at a time
  if (a->a_kmajor && (a->M & 3)) return PTAMD_ERR_BAD_SHAPE;
  if (a->b_kmajor && (a->N & 3)) return PTAMD_ERR_BAD_SHAPE;
  if (!pt_aligned16(a->A) || !pt_aligned16(a->B)) return PTAMD_ERR_ALIGN;
  if (a->dropout_p < 0.f || a->dropout_p >= 1.f) return PTAMD_ERR_BAD_SHAPE;
  splits = a->split_k > 1 ? a->split_k : 1;
  constexpr int BK = 32;
  const int kblocks = (a->K + BK - 1) / BK;
  if (splits > kblocks) splits = kblocks;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.A = a->A; p.lda = a->lda; p.B = a->B; p.ldb = a->ldb; p.C = a->C; p.ldc = a->ldc;
  p.bias = a->bias; p.residual = a->residual; p.ldr = a->ldr; p.flags = a->flags & ~PTAMD_EPI_SLABS;
  // slabs left to the caller: nothing of an epilogue may hang on the product
  if ((a->flags & PTAMD_EPI_SLABS) && (a->bias || a->residual || a->colsum || a->gate_mask || a->dropout_p != 0.f ||
                                      (a->flags & (PTAMD_EPI_RELU | PTAMD_EPI_TANH | PTAMD_EPI_ACCUM | PTAMD_EPI_GATE))))
    return PTAMD_ERR_BAD_SHAPE;
  p.dropout_p = a->dropout_p; p.seed = a->seed; p.stream_id = a->stream_id; p.gate_scale = a->gate_scale;
  p.reserved_cus = a->reserved_cus;
  p.k_per_split = ((kblocks + splits - 1) / splits) * BK;
  splits = (a->K + p.k_per_split - 1) / p.k_per_split;
  p.splits = splits;
  p.vec_epilogue = !(a->N & 3) && !(a->ldc & 3) && pt_aligned16(a->C) &&
                   (!a->residual || (!(a->ldr & 3) && pt_aligned16(a->residual))) &&
                   (splits == 1 || pt_aligned16(a->workspace));
  p.slab = 0;
  p.colsum = a->colsum;
  p.colsum_share = 1;
  {
    const int tiles_n = (a->N + BN - 1) / BN;
    while (p.colsum_share * 2 <= tiles_n && p.colsum_share < 16) p.colsum_share *= 2;
  }
  if (a->colsum && !a->a_kmajor) return PTAMD_ERR_BAD_SHAPE;
  if ((a->flags & PTAMD_EPI_GATE) && (a->residual == nullptr) == (a->gate_mask == nullptr)) return PTAMD_ERR_BAD_SHAPE;
  if (a->gate_mask && !(a->flags & PTAMD_EPI_GATE)) return PTAMD_ERR_BAD_SHAPE;
  p.gate_mask = a->gate_mask; p.gate_mask_out = nullptr;
  p.mask_rb = (a->M + 31) / 32; p.mask_cb = (a->N + 31) / 32;
  user_c = a->C;
  if (splits > 1) {
    if (!a->workspace || a->workspace_bytes < slab_bytes(a->M, a->N, splits)) return PTAMD_ERR_WORKSPACE;
    p.slab = (size_t)a->M * a->N;
    p.C = static_cast<float *>(a->workspace);
    if (a->colsum) p.colsum = p.C + (size_t)splits * p.slab;
  }
  mode = resolve_mode(a);
  // the 1-bit gate is read in the float4 epilogue of the f16x2 kernels, in unsplit products
  if (a->gate_mask && (mode != PTAMD_GEMM_F16X2 || !p.vec_epilogue || splits > 1 || a->dropout_p != 0.f)) return PTAMD_ERR_BAD_SHAPE;
  p.scale_a = p.scale_b = nullptr;
  p.scale_a_stride = p.scale_b_stride = 1;
  if (mode == PTAMD_GEMM_F16X2) {
    if (a->a_scale && a->a_scale_stride != 0 && a->a_scale_stride != 1) return PTAMD_ERR_BAD_SHAPE;
    if (a->b_scale && a->b_scale_stride != 0 && a->b_scale_stride != 1) return PTAMD_ERR_BAD_SHAPE;
  }
  return PTAMD_OK;
}
}  // namespace

size_t ptamd_gate_mask_bytes(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  return (size_t)((M + 31) / 32) * ((N + 31) / 32) * 16 * sizeof(uint64_t);
}

int ptamd_gemm(const ptamd_gemm_args *a, void *stream) {
  GemmParams p;
  int splits, mode;
  float *user_c;
  if (const int rc = build_params(a, p, splits, mode, user_c)) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (mode == PTAMD_GEMM_F16X2) {
    // row scales: the caller's (a_scale / b_scale: written by the kernels that produced the operands, or bounds of the row
    // maxima) or, for an operand that comes without, a pass over it here; those live behind the split-K slabs
    uint32_t *sa = nullptr, *sb = nullptr;
    if (!a->a_scale || !a->b_scale) {
      if (!a->workspace || !pt_aligned16(a->workspace) || a->workspace_bytes < ptamd_gemm_workspace_bytes(a->M, a->N, splits))
        return PTAMD_ERR_WORKSPACE;
      uint32_t *ws = reinterpret_cast<uint32_t *>(static_cast<char *>(a->workspace) + slab_bytes(a->M, a->N, splits));
      if (!a->a_scale) sa = ws;
      if (!a->b_scale) sb = ws + round4(a->M);
      if (const int rs = launch_row_scales(p, a->a_kmajor != 0, a->b_kmajor != 0, sa, sb, st)) return rs;
    }
    p.scale_a = a->a_scale ? a->a_scale : sa;
    p.scale_b = a->b_scale ? a->b_scale : sb;
    p.scale_a_stride = a->a_scale ? a->a_scale_stride : 1;
    p.scale_b_stride = a->b_scale ? a->b_scale_stride : 1;
  }
  const int products = mode == PTAMD_GEMM_BF16X3_FULL ? 9 : mode == PTAMD_GEMM_F16X2 ? 3 : 6;
  const int rc = mode == PTAMD_GEMM_F32 ? launch_f32(p, a->a_kmajor != 0, a->b_kmajor != 0, splits, st)
                                        : launch_split(p, a->a_kmajor != 0, a->b_kmajor != 0, splits, products, st);
  if (rc || splits == 1 || (a->flags & PTAMD_EPI_SLABS)) return rc;   // (PTAMD_EPI_SLABS: the caller sums [splits][M][N] in `workspace`)
  const float *slabs = p.C;
  p.C = user_c;
  const float *cs_slabs = a->colsum ? slabs + (size_t)splits * p.slab : nullptr;
  return launch_splitk_reduce(p, slabs, splits, cs_slabs, a->colsum, st);
}

int ptamd_gemm_group(const ptamd_gemm_args *args, int n, void *stream) {
  if (!args || n <= 0 || n > MAX_GROUP) return PTAMD_ERR_BAD_SHAPE;
  GemmGroup g;
  ReduceGroup r;
  g.n = r.n = n;
  int items = 0, blocks = 0;
  for (int j = 0; j < n; ++j) {
    const ptamd_gemm_args *a = args + j;
    GemmParams &p = g.p[j];
    int splits, mode;
    float *user_c;
    if (const int rc = build_params(a, p, splits, mode, user_c)) return rc;
    // what a member has to be: a k-major x k-major product (a weight gradient) in f16x2 arithmetic with the scales of both
    // operands given, written as split-K slabs (so: split_k >= 2) and accumulated into C; a bias gradient in all or in none
    if (!a->a_kmajor || !a->b_kmajor || mode != PTAMD_GEMM_F16X2 || !a->a_scale || !a->b_scale || splits < 2 ||
        a->flags != PTAMD_EPI_ACCUM || a->bias || a->residual || a->gate_mask || a->dropout_p != 0.f || (a->N & 3) || (a->ldc & 3) ||
        !pt_aligned16(a->C) || !pt_aligned16(a->workspace) || (a->colsum != nullptr) != (args[0].colsum != nullptr) ||
        a->reserved_cus != args[0].reserved_cus)
      return PTAMD_ERR_BAD_SHAPE;
    p.scale_a = a->a_scale; p.scale_b = a->b_scale;
    p.scale_a_stride = a->a_scale_stride; p.scale_b_stride = a->b_scale_stride;
    g.first[j] = items;
    items += ((p.M + 255) / 256) * ((p.N + BN - 1) / BN) * splits;
    ReduceMember &m = r.m[j];
    m.p = p;
    m.p.C = user_c;
    m.slabs = p.C;
    m.cs_slabs = a->colsum ? p.C + (size_t)splits * p.slab : nullptr;
    m.colsum = a->colsum;
    m.splits = splits;
    m.first_block = blocks;
    const size_t work = (size_t)p.M * (p.N >> 2), cs_work = a->colsum ? (size_t)p.M * 16 : 0;
    blocks += (int)(((work > cs_work ? work : cs_work) + 255) / 256);
  }
  for (int j = n; j <= MAX_GROUP; ++j) g.first[j] = items;
  for (int j = n; j < MAX_GROUP; ++j) {
    g.p[j] = g.p[0];
    r.m[j] = r.m[0];
    r.m[j].first_block = blocks;
  }
  r.blocks = blocks;
  hipStream_t st = (hipStream_t)stream;
  if (const int rc = launch_group_f16x2(g, st)) return rc;
  return launch_splitk_reduce_group(r, st);
}

}  // extern "C"
