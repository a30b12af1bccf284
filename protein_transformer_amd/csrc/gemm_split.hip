// fp32 GEMM on the CDNA4 bf16 matrix pipe: each f32 operand is split exactly into three bf16 terms while it is
// staged into LDS, and the product is evaluated as 6 (or 9) v_mfma_f32_32x32x16_bf16 products with f32 accumulation.
//
// Same role, interface and epilogues as gemm.hip (every torch.nn.Linear of the reference encoder and the backward
// GEMMs: Attention.py:38-41,49,69; Sublayers.py:28-34; encoder_only.py:18,39-41) - see ptamd_gemm_set_mode.
//
// Why: on MI355X the f32-input MFMA runs at the vector rate (157 TF/s) while the bf16 MFMA is 16x faster.  An f32
// has 24 significand bits = 3 x the 8 of a bf16 and the same exponent range, so with round-to-nearest at each level
//     x = x1 + x2 + x3  exactly,  |x2| <= 2^-8 |x|,  |x3| <= 2^-16 |x|,
// every bf16 x bf16 product is exact in the f32 accumulator, and
//     x*y = x1y1 + (x1y2 + x2y1) + (x1y3 + x2y2 + x3y1) + [x2y3 + x3y2 + x3y3],   [..] <= 2^-23 |x y|.
// Six products cost 6/16 of the f32 pipe time; the dropped bracket is below the rounding error the f32 fma chain
// makes itself over K >= 64 terms (tests/test_gpu_kernels.py compares both with fp64).
//
// Tiling: workgroup = 128 x 128 outputs, 4 wavefronts x (2 x 2) MFMA tiles of 32 x 32, K advances 32 per stage,
// persistent workgroups (2 per CU) over (tile, K-split) items as in gemm.hip.  A stage is fetched as f32 into
// registers (16-byte coalesced loads, issued one stage ahead), split with v_cvt_pk_bf16_f32 and written to a
// single-buffered LDS image of 3 planes per operand (61 KB):
//   K-contiguous operand  -> plane [row][32 + 8 pad] bf16, fragment = one conflict-free ds_read_b128 (8 k of a row)
//   row-contiguous operand -> plane [k][128 + 32 pad] bf16 (no transposition on the way in), fragment = two
//                             ds_read_b64_tr_b16 (the LDS transpose read delivers 4 k of one row per lane).
// The second workgroup of the CU keeps the matrix pipe busy while this one converts and stores a stage.
#include <stdlib.h>

#include "gemm_common.h"

namespace ptgemm {
namespace {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int SBK = 32;              // f32 k per stage
constexpr int LD_RK = 40;            // bf16 per row of a [row][k] plane: 80 B = odd multiple of 16 B
constexpr int LD_KR = 160;           // bf16 per k of a [k][row] plane: 320 B, 4 consecutive k hit 4 different 64-B bank groups
constexpr int PLANE = 128 * LD_RK;   // = SBK * LD_KR = 5120 bf16 per plane per operand
constexpr int OPERAND = 3 * PLANE;
constexpr size_t LDS_BYTES = (size_t)2 * OPERAND * sizeof(unsigned short);  // 61440
static_assert(PLANE == SBK * LD_KR, "both plane layouts must have the same size");

// (x0, x1) -> three packed bf16 pairs with x = t1 + t2 + t3 exactly (round to nearest even at each level)
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t &t1, uint32_t &t2, uint32_t &t3) {
  const f32x2 v = {x0, x1};
  t1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
  const f32x2 r = {x0 - __uint_as_float(t1 << 16), x1 - __uint_as_float(t1 & 0xffff0000u)};
  t2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
  const f32x2 q = {r.x - __uint_as_float(t2 << 16), r.y - __uint_as_float(t2 & 0xffff0000u)};
  t3 = __builtin_bit_cast(uint32_t, __builtin_convertvector(q, bf16x2));
}

// registers of one stage (thread mapping of load_stage<KMAJOR, 32>) -> the three LDS planes of the operand at `s`
template <bool KMAJOR>
__device__ __forceinline__ void store_split(unsigned short *__restrict__ s, int tid, const float4 (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint2 t1, t2, t3;
    split_pair(v[i].x, v[i].y, t1.x, t2.x, t3.x);
    split_pair(v[i].z, v[i].w, t1.y, t2.y, t3.y);
    int off;
    if (!KMAJOR) off = (tid / 8 + 32 * i) * LD_RK + 4 * (tid % 8);   // 4 consecutive k of one row
    else off = ((tid >> 5) + 8 * i) * LD_KR + 4 * (tid & 31);        // 4 consecutive rows of one k
    *reinterpret_cast<uint2 *>(s + off) = t1;
    *reinterpret_cast<uint2 *>(s + PLANE + off) = t2;
    *reinterpret_cast<uint2 *>(s + 2 * PLANE + off) = t3;
  }
}

// MFMA operand of the 32 tile rows starting at r0 for the 16 k of step ks: lane l holds row r0 + (l & 31),
// k = 16 ks + 8 (l >> 5) + 0..7, for each of the three planes
template <bool KMAJOR>
__device__ __forceinline__ void read_frags(const unsigned short *__restrict__ s, int r0, int lane, int ks, bf16x8 (&f)[3]) {
  if (!KMAJOR) {
    const unsigned short *q = s + (r0 + (lane & 31)) * LD_RK + 16 * ks + 8 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < 3; ++t) f[t] = *reinterpret_cast<const bf16x8 *>(q + t * PLANE);
  } else {
    // ds_read_b64_tr_b16: within a 16-lane group, lane q supplies the address of 4 contiguous bf16 = columns
    // 4 (q & 3) .. +3 of matrix row (q >> 2) and receives column q of the 4 rows.  Rows = 4 consecutive k,
    // columns = 16 consecutive tile rows.
    const int q16 = lane & 15;
    const unsigned short *q = s + (16 * ks + 8 * (lane >> 5) + (q16 >> 2)) * LD_KR + r0 + (lane & 16) + 4 * (q16 & 3);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(q + t * PLANE));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(q + t * PLANE + 4 * LD_KR));
      const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      f[t] = __builtin_bit_cast(bf16x8, both);
    }
  }
}

template <bool A_KMAJOR, bool B_KMAJOR, int NPROD>
__global__ __launch_bounds__(NT, 2) void gemm_bf16x3_mfma_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short *const sA = smem, *const sB = smem + OPERAND;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const WorkRange work(p);
  int w = work.begin, bm0, bn0, z;
  if (w >= work.end) return;
  work.decode(w, bm0, bn0, z);
  int kbeg = z * p.k_per_split, kend = min(p.K, kbeg + p.k_per_split);

  float4 ra[4], rb[4];
  load_stage<A_KMAJOR, SBK>(p.A, p.lda, p.M, bm0, kend, kbeg, tid, ra);
  load_stage<B_KMAJOR, SBK>(p.B, p.ldb, p.N, bn0, kend, kbeg, tid, rb);

  const bool partial = p.slab != 0;
  const uint32_t thr = dropout_threshold(p.dropout_p);
  const float keep_scale = 1.f / (1.f - p.dropout_p);

  // the (A term, B term) pairs, smallest products first
  constexpr int PA[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}, PB[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};

  while (true) {
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Bias gradient = column sums of the k-major A operand, taken from the f32 registers of the stage loader.  With
    // split-K slabs the N tiles of one (M tile, split) share the work: N tile tn takes every cs_share-th k of a
    // stage starting at tn; without slabs the first N tile does it alone and accumulates in place.
    const int cs_share = partial ? p.colsum_share : 1, cs_first = (bn0 / BN) & (cs_share - 1);
    const bool do_colsum = A_KMAJOR && p.colsum != nullptr && (partial ? bn0 / BN < cs_share : bn0 == 0);
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);

    const int w_next = w + 1;
    const bool has_next = w_next < work.end;
    int nbm0 = 0, nbn0 = 0, nz = 0, nkbeg = 0, nkend = 0;
    if (has_next) {
      work.decode(w_next, nbm0, nbn0, nz);
      nkbeg = nz * p.k_per_split;
      nkend = min(p.K, nkbeg + p.k_per_split);
    }

    for (int k0 = kbeg; k0 < kend; k0 += SBK) {
      if (A_KMAJOR && do_colsum) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if ((((tid >> 5) + 8 * i) & (cs_share - 1)) == cs_first) {
            csum.x += ra[i].x; csum.y += ra[i].y; csum.z += ra[i].z; csum.w += ra[i].w;
          }
      }
      store_split<A_KMAJOR>(sA, tid, ra);
      store_split<B_KMAJOR>(sB, tid, rb);
      __syncthreads();
      if (k0 + SBK < kend) {  // next stage of this item, in flight under this stage's MFMAs
        load_stage<A_KMAJOR, SBK>(p.A, p.lda, p.M, bm0, kend, k0 + SBK, tid, ra);
        load_stage<B_KMAJOR, SBK>(p.B, p.ldb, p.N, bn0, kend, k0 + SBK, tid, rb);
      } else if (has_next) {  // first stage of the next item: flies under this item's last stage + epilogue
        load_stage<A_KMAJOR, SBK>(p.A, p.lda, p.M, nbm0, nkend, nkbeg, tid, ra);
        load_stage<B_KMAJOR, SBK>(p.B, p.ldb, p.N, nbn0, nkend, nkbeg, tid, rb);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 fa[2][3], fb[2][3];
        read_frags<A_KMAJOR>(sA, wm * 64, lane, ks, fa[0]);
        read_frags<A_KMAJOR>(sA, wm * 64 + 32, lane, ks, fa[1]);
        read_frags<B_KMAJOR>(sB, wn * 64, lane, ks, fb[0]);
        read_frags<B_KMAJOR>(sB, wn * 64 + 32, lane, ks, fb[1]);
#pragma unroll
        for (int t = 9 - NPROD; t < 9; ++t) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][PA[t]], fb[0][PB[t]], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][PA[t]], fb[1][PB[t]], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][PA[t]], fb[0][PB[t]], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][PA[t]], fb[1][PB[t]], acc[1][1], 0, 0, 0);
        }
      }
      __syncthreads();  // every wavefront is done with this stage's LDS image
    }

    float *C = p.C + (partial ? (size_t)z * p.slab : 0);
    const int ldc = partial ? p.N : p.ldc;
    float *const fsm = reinterpret_cast<float *>(smem);
    if (p.vec_epilogue) {
      tile_epilogue_vec(p, acc, C, ldc, partial, bm0, bn0, wm, wn, lane, thr, keep_scale, fsm + wave * 2048);
      __syncthreads();  // the LDS image is reused by the column sums / the next item's first stage
    } else {
      tile_epilogue_scalar(p, acc, C, ldc, partial, bm0, bn0, wm, wn, lane, thr, keep_scale);
    }
    if (A_KMAJOR && do_colsum) {  // block-uniform: add up the 8 k-groups of the loader
      reinterpret_cast<float4 *>(fsm)[tid] = csum;  // [k group = tid >> 5][row quad = tid & 31]
      __syncthreads();
      if (tid < 128 && bm0 + tid < p.M) {
        float tot = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) tot += fsm[g * 128 + tid];
        if (partial) p.colsum[((size_t)z * cs_share + cs_first) * p.M + bm0 + tid] = tot;
        else p.colsum[bm0 + tid] += tot;
      }
      __syncthreads();
    }
    if (!has_next) break;
    w = w_next; bm0 = nbm0; bn0 = nbn0; z = nz; kbeg = nkbeg; kend = nkend;
  }
}

template <bool AK, bool BKM, int NPROD>
int launch(const GemmParams &p, int splits, hipStream_t st) {
  const int work = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * splits;
  auto kern = gemm_bf16x3_mfma_kernel<AK, BKM, NPROD>;
  static bool attr_set = false;
  if (!attr_set) {
    PT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)LDS_BYTES));
    attr_set = true;
  }
  const int slots = persistent_grid() * 2;
  const int grid = work < slots ? work : slots;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), LDS_BYTES, st, p);
  return pt_check_launch();
}

template <int NPROD>
int launch_layout(const GemmParams &p, bool ak, bool bk, int splits, hipStream_t st) {
  if (!ak && !bk) return launch<false, false, NPROD>(p, splits, st);
  if (!ak && bk) return launch<false, true, NPROD>(p, splits, st);
  if (ak && !bk) return launch<true, false, NPROD>(p, splits, st);
  return launch<true, true, NPROD>(p, splits, st);
}

}  // namespace

int launch_split(const GemmParams &p, bool a_kmajor, bool b_kmajor, int splits, int products, hipStream_t st) {
  return products == 9 ? launch_layout<9>(p, a_kmajor, b_kmajor, splits, st)
                       : launch_layout<6>(p, a_kmajor, b_kmajor, splits, st);
}

}  // namespace ptgemm
