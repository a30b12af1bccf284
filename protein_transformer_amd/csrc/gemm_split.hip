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
// Tiling: workgroup = 128 x 128 outputs, 4 wavefronts x (2 x 2) MFMA tiles of 32 x 32, K advances 16 (one MFMA k
// step) per stage, persistent workgroups (2 per CU) over (tile, K-split) items as in gemm.hip.  A stage is fetched as
// f32 into registers (16-byte coalesced loads, issued TWO stages ahead), split with v_cvt_pk_bf16_f32 and written to
// the other half of a double-buffered LDS image of 3 planes per operand while the MFMAs of the current stage run:
//   K-contiguous operand  -> plane [row][16 + 8 pad] bf16, fragment = one conflict-free ds_read_b128 (8 k of a row)
//   row-contiguous operand -> plane [k][128 + 32 pad] bf16 (no transposition on the way in), fragment = two
//                             ds_read_b64_tr_b16 (the LDS transpose read delivers 4 k of one row per lane).
// One barrier per stage; the stage stream runs across work items, so the first stages of the next tile are fetched,
// converted and stored under the last MFMAs and the epilogue of the current one.
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

constexpr int SBK = 16;              // f32 k per stage = one bf16 MFMA k step
constexpr int LD_RK = 24;            // bf16 per row of a [row][k] plane: 48 B = odd multiple of 16 B
constexpr int LD_KR = 160;           // bf16 per k of a [k][row] plane: 320 B, 4 consecutive k hit 4 different 64-B bank groups
constexpr int PLANE = 128 * LD_RK;   // 3072 bf16 per plane per operand (the [k][row] form needs 16 * 160 = 2560)
constexpr int OPERAND = 3 * PLANE;
constexpr int STAGE = 2 * OPERAND;   // A planes then B planes: 36864 B, also holds the 32 KB epilogue scratch
constexpr size_t LDS_BYTES = (size_t)2 * STAGE * sizeof(unsigned short);  // 73728: two workgroups per CU
static_assert(PLANE >= SBK * LD_KR, "plane must hold either layout");
static_assert(STAGE * sizeof(unsigned short) >= 4 * 2048 * sizeof(float), "epilogue scratch must fit in one stage buffer");

// (x0, x1) -> three packed bf16 pairs with x = t1 + t2 + t3 exactly (round to nearest even at each level)
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t &t1, uint32_t &t2, uint32_t &t3) {
  const f32x2 v = {x0, x1};
  t1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
  const f32x2 r = {x0 - __uint_as_float(t1 << 16), x1 - __uint_as_float(t1 & 0xffff0000u)};
  t2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
  const f32x2 q = {r.x - __uint_as_float(t2 << 16), r.y - __uint_as_float(t2 & 0xffff0000u)};
  t3 = __builtin_bit_cast(uint32_t, __builtin_convertvector(q, bf16x2));
}

// One stage of one operand, global -> registers: 2 float4 per thread.  The loads are UNCONDITIONAL (addresses clamped
// into the operand) and nothing touches the registers until store_split two stages later: a predicated load becomes
// a branch, behind which the compiler can no longer count the loads in flight, and a select on the loaded value
// would wait for it at once - either way the prefetch distance collapses.  Out-of-range ROWS need no masking (they
// only feed output rows that are never stored); out-of-range K is zeroed in store_split.
template <bool KMAJOR>
__device__ __forceinline__ void load_raw(const float *__restrict__ src, int ld, int rows, int r0, int kend, int k0, int tid,
                                         float4 (&v)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (!KMAJOR) {  // src[row][k]: 4 lanes cover the 16 k of one row
      const int row = min(r0 + tid / 4 + 64 * i, rows - 1), k = min(k0 + 4 * (tid % 4), kend - 4);
      v[i] = *reinterpret_cast<const float4 *>(src + (size_t)row * ld + k);
    } else {  // src[k][row]: 32 lanes cover 128 consecutive rows of one k (512 B)
      const int k = min(k0 + (tid >> 5) + 8 * i, kend - 1), row = min(r0 + 4 * (tid & 31), rows - 4);
      v[i] = *reinterpret_cast<const float4 *>(src + (size_t)k * ld + row);
    }
  }
}

// registers of one stage -> the three LDS planes of the operand at `s`; klim = number of valid k in the stage
template <bool KMAJOR>
__device__ __forceinline__ void store_split(unsigned short *__restrict__ s, int tid, const float4 (&v)[2], int klim) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const bool ok = (KMAJOR ? (tid >> 5) + 8 * i : 4 * (tid % 4)) < klim;
    const float4 x = make_float4(ok ? v[i].x : 0.f, ok ? v[i].y : 0.f, ok ? v[i].z : 0.f, ok ? v[i].w : 0.f);
    uint2 t1, t2, t3;
    split_pair(x.x, x.y, t1.x, t2.x, t3.x);
    split_pair(x.z, x.w, t1.y, t2.y, t3.y);
    int off;
    if (!KMAJOR) off = (tid / 4 + 64 * i) * LD_RK + 4 * (tid % 4);   // 4 consecutive k of one row
    else off = ((tid >> 5) + 8 * i) * LD_KR + 4 * (tid & 31);        // 4 consecutive rows of one k
    *reinterpret_cast<uint2 *>(s + off) = t1;
    *reinterpret_cast<uint2 *>(s + PLANE + off) = t2;
    *reinterpret_cast<uint2 *>(s + 2 * PLANE + off) = t3;
  }
}

// MFMA operand of the 32 tile rows starting at r0: lane l holds row r0 + (l & 31), k = 8 (l >> 5) + 0..7 of the
// stage, for each of the three planes
template <bool KMAJOR>
__device__ __forceinline__ void read_frags(const unsigned short *__restrict__ s, int r0, int lane, bf16x8 (&f)[3]) {
  if (!KMAJOR) {
    const unsigned short *q = s + (r0 + (lane & 31)) * LD_RK + 8 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < 3; ++t) f[t] = *reinterpret_cast<const bf16x8 *>(q + t * PLANE);
  } else {
    // ds_read_b64_tr_b16: within a 16-lane group, lane q supplies the address of 4 contiguous bf16 = columns
    // 4 (q & 3) .. +3 of matrix row (q >> 2) and receives column q of the 4 rows.  Rows = 4 consecutive k,
    // columns = 16 consecutive tile rows.
    const int q16 = lane & 15;
    const unsigned short *q = s + (8 * (lane >> 5) + (q16 >> 2)) * LD_KR + r0 + (lane & 16) + 4 * (q16 & 3);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(q + t * PLANE));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(q + t * PLANE + 4 * LD_KR));
      const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      f[t] = __builtin_bit_cast(bf16x8, both);
    }
  }
}

struct Item {  // one (output tile, K split) work item
  int bm0, bn0, z, kbeg, kend;
};

template <bool A_KMAJOR, bool B_KMAJOR, int NPROD>
__global__ __launch_bounds__(NT, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bf16x3_mfma_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const WorkRange work(p);
  if (work.begin >= work.end) return;
  auto item_at = [&](int logical) __attribute__((always_inline)) {
    Item it;
    work.decode(logical, it.bm0, it.bn0, it.z);
    it.kbeg = it.z * p.k_per_split;
    it.kend = min(p.K, it.kbeg + p.k_per_split);
    return it;
  };

  const bool partial = p.slab != 0;
  const uint32_t thr = dropout_threshold(p.dropout_p);
  const float keep_scale = 1.f / (1.f - p.dropout_p);
  // Bias gradient = column sums of the k-major A operand, taken from the f32 registers of the stage loader.  With
  // split-K slabs the N tiles of one (M tile, split) share the work: N tile tn takes every cs_share-th k of a
  // stage starting at tn; without slabs the first N tile does it alone and accumulates in place.
  const int cs_share = partial ? p.colsum_share : 1;
  auto colsum_first = [&](const Item &it) __attribute__((always_inline)) { return (it.bn0 / BN) & (cs_share - 1); };
  auto colsum_on = [&](const Item &it) __attribute__((always_inline)) {
    return A_KMAJOR && p.colsum != nullptr && (partial ? it.bn0 / BN < cs_share : it.bn0 == 0);
  };
  auto colsum_add = [&](const Item &it, const float4 (&v)[2], int klim, float4 &acc4) __attribute__((always_inline)) {
    if (colsum_on(it)) {
      const int first = colsum_first(it);
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if ((((tid >> 5) + 8 * i) & (cs_share - 1)) == first && (tid >> 5) + 8 * i < klim) {
          acc4.x += v[i].x; acc4.y += v[i].y; acc4.z += v[i].z; acc4.w += v[i].w;
        }
    }
  };

  // ---- the stage stream: a load cursor runs two stages ahead of the compute cursor, across work items
  int w = work.begin;
  Item cur = item_at(w);
  bool has_next = w + 1 < work.end;
  Item nxt = has_next ? item_at(w + 1) : cur;
  int lw = w, lk0 = cur.kbeg;
  Item lit = cur;
  // (past the last stage of the range the cursor stays put and the stage is fetched again: the loop body has no
  // branch around its loads and stores, see load_raw)
  auto issue_load = [&](float4 (&a)[2], float4 (&b)[2], int &klim) __attribute__((always_inline)) {
    load_raw<A_KMAJOR>(p.A, p.lda, p.M, lit.bm0, lit.kend, lk0, tid, a);
    load_raw<B_KMAJOR>(p.B, p.ldb, p.N, lit.bn0, lit.kend, lk0, tid, b);
    klim = lit.kend - lk0;
    if (lk0 + SBK < lit.kend) {
      lk0 += SBK;
    } else if (lw + 1 < work.end) {
      lit = item_at(++lw);
      lk0 = lit.kbeg;
    }
  };

  // the (A term, B term) pairs, smallest products first
  constexpr int PA[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}, PB[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};

  f32x16 acc[2][2];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  zero_acc();
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f), csum_next = make_float4(0.f, 0.f, 0.f, 0.f);

  float4 r0a[2], r0b[2], r1a[2], r1b[2];
  int klim0 = 0, klim1 = 0;
  issue_load(r0a, r0b, klim0);
  issue_load(r1a, r1b, klim1);
  colsum_add(cur, r0a, klim0, csum);
  store_split<A_KMAJOR>(smem, tid, r0a, klim0);
  store_split<B_KMAJOR>(smem + OPERAND, tid, r0b, klim0);
  __syncthreads();
  int cb = 0, ck0 = cur.kbeg;
  bool done = false;

  // one stage: fetch stage g+2 into (la, lb); MFMAs of stage g from LDS buffer cb; convert + store stage g+1 (held in
  // (sa, sb)) into the other buffer; barrier; at the end of an item its epilogue.
  auto stage = [&](float4 (&la)[2], float4 (&lb)[2], int &lklim, const float4 (&sa)[2], const float4 (&sb)[2],
                   int sklim) __attribute__((always_inline)) {
    issue_load(la, lb, lklim);
    const unsigned short *bufA = smem + cb * STAGE, *bufB = bufA + OPERAND;
    unsigned short *othA = smem + (cb ^ 1) * STAGE, *othB = othA + OPERAND;
    bf16x8 fa[2][3], fb[2][3];
    read_frags<A_KMAJOR>(bufA, wm * 64, lane, fa[0]);
    read_frags<A_KMAJOR>(bufA, wm * 64 + 32, lane, fa[1]);
    read_frags<B_KMAJOR>(bufB, wn * 64, lane, fb[0]);
    read_frags<B_KMAJOR>(bufB, wn * 64 + 32, lane, fb[1]);
#pragma unroll
    for (int t = 9 - NPROD; t < 9; ++t) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][PA[t]], fb[0][PB[t]], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][PA[t]], fb[1][PB[t]], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][PA[t]], fb[0][PB[t]], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][PA[t]], fb[1][PB[t]], acc[1][1], 0, 0, 0);
    }
    const bool last = ck0 + SBK >= cur.kend;
    // stage g+1 belongs to this item or is the first stage of the next one (after the very last stage of the range
    // the registers hold a re-fetched stage: stored, never read)
    if (A_KMAJOR) {
      if (!last) colsum_add(cur, sa, sklim, csum);
      else if (has_next) colsum_add(nxt, sa, sklim, csum_next);
    }
    store_split<A_KMAJOR>(othA, tid, sa, sklim);
    store_split<B_KMAJOR>(othB, tid, sb, sklim);
    __syncthreads();  // stage g+1 is complete in LDS; nobody reads buffer cb any more
    ck0 += SBK;
    if (last) {
      float *C = p.C + (partial ? (size_t)cur.z * p.slab : 0);
      const int ldc = partial ? p.N : p.ldc;
      float *const fsm = reinterpret_cast<float *>(smem + cb * STAGE);  // the buffer this stage just released
      if (p.vec_epilogue) {
        tile_epilogue_vec(p, acc, C, ldc, partial, cur.bm0, cur.bn0, wm, wn, lane, thr, keep_scale, fsm + wave * 2048);
      } else {
        tile_epilogue_scalar(p, acc, C, ldc, partial, cur.bm0, cur.bn0, wm, wn, lane, thr, keep_scale);
      }
      if (colsum_on(cur)) {  // block-uniform: add up the 8 k-groups of the loader
        if (p.vec_epilogue) __syncthreads();
        reinterpret_cast<float4 *>(fsm)[tid] = csum;  // [k group = tid >> 5][row quad = tid & 31]
        __syncthreads();
        if (tid < 128 && cur.bm0 + tid < p.M) {
          float tot = 0.f;
#pragma unroll
          for (int g = 0; g < 8; ++g) tot += fsm[g * 128 + tid];
          if (partial) p.colsum[((size_t)cur.z * cs_share + colsum_first(cur)) * p.M + cur.bm0 + tid] = tot;
          else p.colsum[cur.bm0 + tid] += tot;
        }
      }
      if (!has_next) {
        done = true;
      } else {
        __syncthreads();  // the released buffer (epilogue scratch) receives stage g+2 in the next stage
        cur = nxt;
        ++w;
        has_next = w + 1 < work.end;
        if (has_next) nxt = item_at(w + 1);
        ck0 = cur.kbeg;
        csum = csum_next;
        csum_next = make_float4(0.f, 0.f, 0.f, 0.f);
        zero_acc();
      }
    }
    cb ^= 1;
  };

  while (true) {
    stage(r0a, r0b, klim0, r1a, r1b, klim1);
    if (done) break;
    stage(r1a, r1b, klim1, r0a, r0b, klim0);
    if (done) break;
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
