// The split-arithmetic GEMM kernel template and its launchers; instantiated by gemm_split.hip (three bf16 terms,
// 6 / 9 products) and gemm_f16x2.hip (two row-scaled f16 terms, 3 products) - two translation units so that the
// 28 instantiations compile in parallel.
//
// fp32 GEMM on the CDNA4 bf16 / f16 matrix pipe: each f32 operand is split while it is staged into LDS - exactly into
// three bf16 terms (NPROD = 6 or 9 v_mfma_f32_32x32x16_bf16 products per fp32 product), or, after scaling each operand
// row by a power of two, into two f16 terms (NPROD = 3 v_mfma_f32_32x32x16_f16 products; the error model and the
// row-scale pass are described in include/ptamd.h and gemm_f16x2.hip) - with f32 accumulation.  The text below
// describes the bf16 form; the f16 form differs by two planes instead of three, the scale loaded with every stage
// and the two inverse scales applied to the accumulators in front of the epilogue.
//
// Same role, interface and epilogues as gemm.hip (every torch.nn.Linear of the reference encoder and the backward
// GEMMs: Attention.py:38-41,49,69; Sublayers.py:28-34; encoder_only.py:18,39-41) - see the `arith` field of ptamd_gemm_args.
//
// Why: on MI355X the f32-input MFMA runs at the vector rate (157 TF/s) while the bf16 MFMA is 16x faster.  An f32
// has 24 significand bits = 3 x the 8 of a bf16 and the same exponent range, so with round-to-nearest at each level
//     x = x1 + x2 + x3  exactly,  |x2| <= 2^-8 |x|,  |x3| <= 2^-16 |x|,
// every bf16 x bf16 product is exact in the f32 accumulator, and
//     x*y = x1y1 + (x1y2 + x2y1) + (x1y3 + x2y2 + x3y1) + [x2y3 + x3y2 + x3y3],   [..] <= 2^-23 |x y|.
// Six products cost 6/16 of the f32 pipe time; the dropped bracket is below the rounding error the f32 fma chain
// makes itself over K >= 64 terms (tests/test_gpu_kernels.py compares both with fp64).
//
// Structure (MI355X-first): one workgroup per CU = 8 wavefronts, two per SIMD with different jobs.
//   * wavefronts 0-3 are CONSUMERS: each owns a 128 x 64 block (4 x 2 MFMA tiles of 32 x 32, 128 accumulator VGPRs) of
//     the workgroup's 256 x 128 output tile and does nothing but ds_read fragments and issue MFMAs (48 per K step
//     of 16), then the epilogue of the tile through its own LDS scratch;
//   * wavefronts 4-7 are PRODUCERS: they fetch the f32 operands (16-byte coalesced loads from a scalar stage base + a
//     32-bit lane offset, NSETS = 4 stages in flight, unconditional), split them with v_cvt_pk_bf16_f32 and scalar
//     subtractions and write the three bf16 planes of the next stage into the other half of a double-buffered LDS
//     image; they also add up the fused bias gradient (no branch, no memory access and no packed f32 instruction
//     in their loop: each of the three was measured to cost 10-30 %).
//   The matrix pipe of a SIMD is fed by its consumer while its producer uses the VALU, the LDS write path and the
//   vector memory path: the two instruction streams overlap because they belong to different wavefronts.  One
//   s_barrier per stage hands a finished buffer from the producers to the consumers and a consumed one back.
// The 256 x 128 tile (rather than 128 x 128) cuts the L2 -> CU traffic and the LDS write traffic per MFMA by a
// quarter - with the 16x faster pipe both are first-order costs (LDS writes run at ~80 B/clk/CU).
// LDS image of a stage, per operand and plane:
//   K-contiguous operand   -> [row][16 + 8 pad] bf16, fragment = one conflict-free ds_read_b128 (8 k of a row)
//   row-contiguous operand -> [k][rows + 32 pad] bf16 (no transposition on the way in), fragment = two
//                             ds_read_b64_tr_b16 (the LDS transpose read delivers 4 k of one row per lane).
// Workgroups are persistent and walk contiguous (tile, K-split) ranges as in gemm.hip; the stage stream runs across
// work items, so the producers fetch and convert the first stages of the next tile under the epilogue of this one.
#pragma once
#include <type_traits>

#include <stdlib.h>

#include "gemm_common.h"
#include "split_bf16.h"

namespace ptgemm {
namespace {

using namespace ptsplit;  // bf16x8, split_pair (x = t1 + t2 + t3 exactly, scalar subtractions), LDS transpose-read types

constexpr int TBN = 128;             // output tile of a workgroup: (64 TI) x 128, TI = 4 (256 rows) or 2 (128 rows, below)
constexpr int NTHREADS = 512, NPRODUCER = 256;
constexpr int KR_PAD = 32;           // a [k][rows + 32] plane: 4 consecutive k hit 4 different 64-B bank groups
constexpr int SCRATCH_FLOATS = 4 * 2048;                      // epilogue transpose scratch of the 4 consumers
constexpr int COLSUM_AREAS = 3;                               // see the producers' publish / the consumers' read below
constexpr int COLSUM_FLOATS = COLSUM_AREAS * 4 * 256;         // rotating [4 k groups][256 rows] partial sums (256-row tiles only)
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// Geometry of a stage by arithmetic.
//   three bf16 planes (NPROD 6 / 9): 16 k per stage, K-contiguous planes [row][16 + 8 pad] (48-byte rows: conflict-free
//     ds_read_b128), four register sets of global loads in flight;
//   two f16 planes (NPROD 3): 32 k per stage - ONE barrier and one round of the producer / consumer hand-off per 32 k
//     instead of per 16 (the barrier structure cost 0.28 us of a 0.93-us 16-k stage, profiles/r02/r02_gemm_occupancy.txt), whole
//     128-byte lines per K-contiguous operand row and load (16 k touched one 64-byte half of every line).  Two planes of
//     32 k only fit LDS unpadded: K-contiguous planes are [row][32] f16 = four 16-byte chunks per row, chunk c of row r
//     stored at c ^ ((r >> 2) & 3) - any 16 rows x one chunk column cover 16 different 16-byte slots of the 256-byte bank
//     row, so the fragment reads stay conflict-free; two register sets of twice the size keep the same 64 k in flight.
// TI = 32-row MFMA tiles per consumer: 4 = the 256 x 128 workgroup tile; 2 = a 128 x 128 tile for products whose 256-row
// tiles would leave most of the chip idle (few tokens: small batches, the per-GPU share of a strongly scaled batch) - the
// latency floor of such a product is ONE tile's time, which halves with the tile (K-contiguous A only: the weight-gradient
// products split their long K instead).
template <int NPROD, int TI = 4>
struct Geo {
  static constexpr int TBM = 64 * TI;
  static constexpr bool F16 = NPROD == 3;
  static constexpr int NPLANES = F16 ? 2 : 3;
  static constexpr int SBK = F16 ? 32 : 16;              // f32 k per stage
  static constexpr int QK = SBK / 4;                     // float4 per K-contiguous operand row and stage
#if defined(PT_F16_NSETS_TI2)   // measurement build: deeper prefetch of the producers for the 128-row tiles of small batches
  static constexpr int NSETS = F16 ? (TI == 4 ? 2 : PT_F16_NSETS_TI2) : 4;
#else
  static constexpr int NSETS = F16 ? 2 : 4;              // register sets of a producer = stages of global loads in flight (even)
#endif
  static constexpr int LD_RK = F16 ? SBK : SBK + 8;      // f16 / bf16 per row of a [row][k] plane
  static constexpr bool SWZ = F16;                       // chunk swizzle instead of padding
  static constexpr int PLANE_A = cmax(TBM * LD_RK, SBK * (TBM + KR_PAD));
  static constexpr int PLANE_B = cmax(TBN * LD_RK, SBK * (TBN + KR_PAD));
  static constexpr int STAGE = NPLANES * (PLANE_A + PLANE_B);
  static constexpr size_t LDS_BYTES = (size_t)2 * STAGE * sizeof(unsigned short) + (SCRATCH_FLOATS + COLSUM_FLOATS) * sizeof(float);
  static constexpr int NVA = TBM * SBK / (4 * NPRODUCER), NVB = TBN * SBK / (4 * NPRODUCER);   // float4 per producer thread and stage
  template <int ROWS> static constexpr int nv() { return ROWS * SBK / (4 * NPRODUCER); }
  template <int ROWS> static constexpr int plane() { return ROWS == TBM ? PLANE_A : PLANE_B; }
  // element offset of (row, k) in a [row][k] plane, k multiple of 4
  static __device__ __forceinline__ int rk_offset(int row, int k) {
    return SWZ ? row * LD_RK + ((((k >> 3) ^ (row >> 2)) & 3) << 3) + (k & 7) : row * LD_RK + k;
  }
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget of a CU");
};

// One stage of one operand (ROWS tile rows x 16 k), global -> registers of the 256 producer threads: ROWS / 64
// float4 per thread, each from  (scalar base of the stage) + (32-bit per-thread byte offset of the work item),
// so a stage costs no address arithmetic in the vector unit.  The loads are UNCONDITIONAL and nothing touches the
// registers until store_split two stages later: a predicated load becomes a branch, behind which the compiler can
// no longer count the loads in flight, and a select on the loaded value would wait for it at once - either way
// the prefetch distance collapses.  Rows beyond the operand are clamped (they only feed output rows that are never
// stored); a K tail is handled by starting the last stage of an item 16 k before its end and zeroing the k that
// were already consumed (store_split's `kskip`).
// Geometry helpers: G = Geo<NPROD>.  A K-contiguous operand row holds G::QK float4 of a stage; thread pt takes k quad
// pt % QK of the rows pt / QK + (256 / QK) i.  A row-contiguous operand: thread pt takes the row quad pt % LPK of the k
// rows pt / LPK + (256 / LPK) i.
template <typename G, bool KMAJOR, int ROWS>
__device__ __forceinline__ void item_offsets(int ld, int rows, int r0, int pt, uint32_t (&voff)[G::template nv<ROWS>()]) {
  constexpr int LPK = ROWS / 4, NV = G::template nv<ROWS>();  // lanes per k of a row-contiguous operand
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (!KMAJOR) voff[i] = ((uint32_t)min(r0 + pt / G::QK + (NPRODUCER / G::QK) * i, rows - 1) * (uint32_t)ld + 4 * (pt % G::QK)) * 4u;
    else voff[i] = ((uint32_t)(pt / LPK + (NPRODUCER / LPK) * i) * (uint32_t)ld + min(r0 + 4 * (pt % LPK), rows - 4)) * 4u;
  }
}
template <int N>
__device__ __forceinline__ void load_raw(const float *__restrict__ stage_base, const uint32_t (&voff)[N], float4 (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i)
    v[i] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(stage_base) + voff[i]);
}

// the first kskip k of a stage are zeroed (only the last stage of an item whose K range is not a multiple of the stage)
template <typename G, bool KMAJOR, int ROWS>
__device__ __forceinline__ void mask_tail(int pt, float4 (&v)[G::template nv<ROWS>()], int kskip) {
  constexpr int LPK = ROWS / 4, NV = G::template nv<ROWS>();
  if (kskip > 0) {  // uniform
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int kl = KMAJOR ? pt / LPK + (NPRODUCER / LPK) * i : 4 * (pt % G::QK);
      if (KMAJOR) {
        if (kl < kskip) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {  // a float4 = 4 consecutive k: kskip is a multiple of 4 (K, k_per_split are)
        if (kl < kskip) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
}
// registers of one stage -> the three LDS planes of the operand at `s`
template <typename G, bool KMAJOR, int ROWS>
__device__ __forceinline__ void store_split(unsigned short *__restrict__ s, int pt, const float4 (&v)[G::template nv<ROWS>()]) {
  constexpr int LPK = ROWS / 4, PLANE = G::template plane<ROWS>(), LD_KR = ROWS + KR_PAD, NV = G::template nv<ROWS>();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int kl = KMAJOR ? pt / LPK + (NPRODUCER / LPK) * i : 4 * (pt % G::QK);
    uint2 t1, t2, t3;
    split_pair(v[i].x, v[i].y, t1.x, t2.x, t3.x);
    split_pair(v[i].z, v[i].w, t1.y, t2.y, t3.y);
    const int off = KMAJOR ? kl * LD_KR + 4 * (pt % LPK)                                   // 4 consecutive rows of one k
                           : G::rk_offset(pt / G::QK + (NPRODUCER / G::QK) * i, kl);       // 4 consecutive k of one row
    *reinterpret_cast<uint2 *>(s + off) = t1;
    *reinterpret_cast<uint2 *>(s + PLANE + off) = t2;
    *reinterpret_cast<uint2 *>(s + 2 * PLANE + off) = t3;
  }
}

// ---- f16x2 arithmetic (NPROD == 3): every operand row carries a power-of-two scale (gemm_row_scale_kernel below);
// the producers load it with the stage (unconditionally, like the operand itself), multiply and split into TWO f16
// planes, and the consumers undo the two scales on the accumulators before the epilogue.
constexpr int MAX_NV = 8;   // float4 (and, for a K-contiguous operand, row scales) per producer thread and stage
template <typename G, bool KMAJOR, int ROWS>
__device__ __forceinline__ void scale_offsets(int rows, int r0, int pt, uint32_t (&soff)[MAX_NV], int stride = 1) {
  constexpr int LPK = ROWS / 4, NV = G::template nv<ROWS>();
#pragma unroll
  for (int i = 0; i < MAX_NV; ++i) {
    if (!KMAJOR) soff[i] = (uint32_t)min(r0 + pt / G::QK + (NPRODUCER / G::QK) * (i < NV ? i : 0), rows - 1) * 4u * (uint32_t)stride;  // the row of v[i]
    else soff[i] = (uint32_t)min(r0 + 4 * (pt % LPK), rows - 4) * 4u * (uint32_t)stride;              // the 4 rows of every v[i]
  }
}
template <typename G, bool KMAJOR, int ROWS>
__device__ __forceinline__ void load_scales(const uint32_t *__restrict__ scale, const uint32_t (&soff)[MAX_NV], float (&sc)[MAX_NV]) {
  const char *base = reinterpret_cast<const char *>(scale);
  constexpr int NV = G::template nv<ROWS>();
  if (!KMAJOR) {
#pragma unroll
    for (int i = 0; i < NV; ++i) sc[i] = *reinterpret_cast<const float *>(base + soff[i]);
  } else {
    const float4 v = *reinterpret_cast<const float4 *>(base + soff[0]);
    sc[0] = v.x; sc[1] = v.y; sc[2] = v.z; sc[3] = v.w;
  }
}
// float4 number i of the stage (i is a constant after unrolling)
template <typename G, bool KMAJOR, int ROWS>
__device__ __forceinline__ void store_split_f16_one(unsigned short *__restrict__ s, int pt, const float4 &v, const float (&sc)[MAX_NV], int i) {
  constexpr int LPK = ROWS / 4, PLANE = G::template plane<ROWS>(), LD_KR = ROWS + KR_PAD;
  const int kl = KMAJOR ? pt / LPK + (NPRODUCER / LPK) * i : 4 * (pt % G::QK);
  uint2 t1, t2;
  split_quad_f16(v.x, v.y, v.z, v.w, KMAJOR ? sc[0] : sc[i], KMAJOR ? sc[1] : sc[i], KMAJOR ? sc[2] : sc[i], KMAJOR ? sc[3] : sc[i], t1, t2);
  const int off = KMAJOR ? kl * LD_KR + 4 * (pt % LPK) : G::rk_offset(pt / G::QK + (NPRODUCER / G::QK) * i, kl);
  *reinterpret_cast<uint2 *>(s + off) = t1;
  *reinterpret_cast<uint2 *>(s + PLANE + off) = t2;
}
template <typename G, bool KMAJOR, int ROWS>
__device__ __forceinline__ void store_split_f16(unsigned short *__restrict__ s, int pt, const float4 (&v)[G::template nv<ROWS>()],
                                                const float (&sc)[MAX_NV]) {
  constexpr int NV = G::template nv<ROWS>();
#pragma unroll
  for (int i = 0; i < NV; ++i) store_split_f16_one<G, KMAJOR, ROWS>(s, pt, v[i], sc, i);
}
// scale (bits of a power of two) of a row whose largest |x| has the bits `amax`: max |x| * scale in [2^14, 2^15);
// rows of zeros / subnormals get the largest finite power.  A larger maximum gives a SMALLER scale (atomicMin).
__device__ __forceinline__ uint32_t row_scale_bits(uint32_t amax) { return min(268u - (amax >> 23), 254u) << 23; }
__device__ __forceinline__ float inverse_scale(uint32_t scale_bits) { return __uint_as_float((254u << 23) - scale_bits); }

// MFMA operand of the 32 tile rows starting at r0, plane t, k step ks of the stage: lane l holds row r0 + (l & 31),
// k = 16 ks + 8 (l >> 5) + 0..7
template <typename G, bool KMAJOR, int ROWS>
__device__ __forceinline__ bf16x8 read_frag(const unsigned short *__restrict__ s, int r0, int lane, int t, int ks) {
  constexpr int PLANE = G::template plane<ROWS>(), LD_KR = ROWS + KR_PAD;
  if (!KMAJOR) {
    return *reinterpret_cast<const bf16x8 *>(s + t * PLANE + G::rk_offset(r0 + (lane & 31), 16 * ks + 8 * (lane >> 5)));
  } else {
    // ds_read_b64_tr_b16: within a 16-lane group, lane q supplies the address of 4 contiguous bf16 = columns
    // 4 (q & 3) .. +3 of matrix row (q >> 2) and receives column q of the 4 rows.  Rows = 4 consecutive k,
    // columns = 16 consecutive tile rows.
    const int q16 = lane & 15;
    const unsigned short *q = s + t * PLANE + (16 * ks + 8 * (lane >> 5) + (q16 >> 2)) * LD_KR + r0 + (lane & 16) + 4 * (q16 & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)q);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(q + 4 * LD_KR));
    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
  }
}

struct Item {  // one (output tile, K split) work item; pj = the member of a group launch it belongs to (0 otherwise)
  int bm0, bn0, z, kbeg, kend, pj;
};

// Cursor over the stage stream of a workgroup: the stages (16 k each) of its work items, in order.
struct Cursor {
  int w, k0;
  Item it;
  bool end;  // set once the cursor was asked to step past the last stage (it then stays on that stage)
};

// GROUPED: the argument is a GemmGroup and every work item names its member; the parameters of the product are then those
// of the member of the item a ROLE is at (the producers run ahead of the consumers and may be in another member).
template <bool GROUPED>
struct KernelArg {
  typedef GemmParams type;
};
template <>
struct KernelArg<true> {
  typedef GemmGroup type;
};
template <bool A_KMAJOR, bool B_KMAJOR, int NPROD, int EPI, int TI = 4, bool GROUPED = false>
__global__ __launch_bounds__(NTHREADS, 1) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bf16x3_mfma_kernel(
    const typename KernelArg<GROUPED>::type arg) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  using G = Geo<NPROD, TI>;
  constexpr int TBM = G::TBM;
  static_assert(TI == 4 || !A_KMAJOR, "the fused bias gradient of k-major A assumes 256-row tiles");
  constexpr int SBK = G::SBK, NSETS = G::NSETS, STAGE = G::STAGE, PLANE_A = G::PLANE_A, NVA = G::NVA, NVB = G::NVB;
  float *const scratch = reinterpret_cast<float *>(smem + 2 * STAGE);
  float *const cs_area = scratch + SCRATCH_FLOATS;

  constexpr bool F16 = NPROD == 3;  // two scaled f16 terms and three products instead of three bf16 terms and six
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto P = [&](int j) __attribute__((always_inline)) -> const GemmParams & {
    if constexpr (GROUPED) return arg.p[j];
    else return arg;
  };
  typename std::conditional<GROUPED, GroupRange, WorkRange>::type work(arg, TBM, TBN);
  if (work.begin >= work.end) return;
  auto item_at = [&](int logical) __attribute__((always_inline)) {
    Item it;
    if constexpr (GROUPED) work.decode(arg, logical, it.pj, it.bm0, it.bn0, it.z);
    else {
      work.decode(logical, it.bm0, it.bn0, it.z);
      it.pj = 0;
    }
    it.kbeg = it.z * P(it.pj).k_per_split;
    it.kend = min(P(it.pj).K, it.kbeg + P(it.pj).k_per_split);
    return it;
  };
  // one step of a cursor; past the last stage of the range it stays on that stage
  auto advance = [&](Cursor &c) __attribute__((always_inline)) {
    if (c.k0 + SBK < c.it.kend) {
      c.k0 += SBK;
    } else if (c.w + 1 < work.end) {
      c.it = item_at(++c.w);
      c.k0 = c.it.kbeg;
    } else {
      c.end = true;
    }
  };
  int total_stages = 0;  // barriers must match between the two roles: both count the stages of the range
  for (int w = work.begin; w < work.end; ++w) {
    const Item it = item_at(w);
    total_stages += (it.kend - it.kbeg + SBK - 1) / SBK;
  }
  const int padded_stages = (total_stages + NSETS - 1) / NSETS * NSETS;  // both roles run this many barriers (+1)
  // (a group: every member writes slabs, and its members agree on having a bias gradient - ptamd_gemm_group checks both)
  const bool partial = P(0).slab != 0;
  // Bias gradient = column sums of the k-major A operand.  The producers add up the f32 registers of their loader
  // (weights 0 / 1 per k row, no branch and no memory access in their loop) and publish one partial row per k group in
  // LDS at the end of an item; the consumers write it out with the tile.  With split-K slabs the N tiles of one
  // (M tile, split) share the work: N tile tn takes every cs_share-th k of a stage starting at tn; without slabs the
  // first N tile does it alone and accumulates in place.
  const bool has_colsum = A_KMAJOR && P(0).colsum != nullptr;
  auto cs_share_of = [&](const Item &it) __attribute__((always_inline)) { return partial ? P(it.pj).colsum_share : 1; };
  auto colsum_first = [&](const Item &it) __attribute__((always_inline)) { return (it.bn0 / TBN) & (cs_share_of(it) - 1); };
  auto colsum_on = [&](const Item &it) __attribute__((always_inline)) {
    return has_colsum && (partial ? it.bn0 / TBN < cs_share_of(it) : it.bn0 == 0);
  };

  if (wave >= 4) {
    // ================================================================ producers
    const int pt = tid - NPRODUCER;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    float cs_w[NVA];  // weight of this thread's k rows (pt / 64 + 4 i) for the item under `st`
#pragma unroll
    for (int i = 0; i < NVA; ++i) cs_w[i] = 0.f;
    int cs_parity = 0;
    auto colsum_weights = [&](const Item &it) __attribute__((always_inline)) {
      const bool on = colsum_on(it);
      const int first = colsum_first(it), cs_share = cs_share_of(it);
#pragma unroll
      for (int i = 0; i < NVA; ++i) cs_w[i] = (on && ((pt / 64 + 4 * i) & (cs_share - 1)) == first) ? 1.f : 0.f;
    };

    Cursor ld = {work.begin, 0, item_at(work.begin), false};  // next stage to fetch
    ld.k0 = ld.it.kbeg;
    Cursor st = ld;                                    // next stage to convert and store
    colsum_weights(st.it);
    uint32_t voa[NVA], vob[NVB];                       // per-thread byte offsets of the item under `ld`
    // operands of the member the item under `ld` belongs to (a group: reloaded when the cursor enters another member)
    const float *opA = P(ld.it.pj).A, *opB = P(ld.it.pj).B;
    int op_lda = P(ld.it.pj).lda, op_ldb = P(ld.it.pj).ldb;
    item_offsets<G, A_KMAJOR, TBM>(op_lda, P(ld.it.pj).M, ld.it.bm0, pt, voa);
    item_offsets<G, B_KMAJOR, TBN>(op_ldb, P(ld.it.pj).N, ld.it.bn0, pt, vob);
    float4 ra[NSETS][NVA], rb[NSETS][NVB];             // NSETS stages in flight (registers)
    float rsa[NSETS][MAX_NV], rsb[NSETS][MAX_NV];      // f16x2 only: the row scales that go with them
    uint32_t soa[MAX_NV], sob[MAX_NV];
    if (F16) {
      scale_offsets<G, A_KMAJOR, TBM>(P(ld.it.pj).M, ld.it.bm0, pt, soa, P(ld.it.pj).scale_a_stride);
      scale_offsets<G, B_KMAJOR, TBN>(P(ld.it.pj).N, ld.it.bn0, pt, sob, P(ld.it.pj).scale_b_stride);
    }
    int rskip[NSETS];
    // what follows the operand loads of a stage: the row scales (see below), the cursor, the offsets of a new item
    auto fetch_tail = [&](float (&sa_)[MAX_NV], float (&sb_)[MAX_NV]) __attribute__((always_inline)) {
      // The row scales of a register set change with the ITEM only: they are loaded with the first stage that each of the
      // NSETS sets receives from an item and stay in its registers (a stage of K-contiguous operands was 12 operand + 12
      // scale loads per thread, and the producers' load ISSUE - ~35 cycles per instruction with four wavefronts at it -
      // is on the critical path of a stage: profiles/r03/r03_gemm_stage_trace.txt).  Uniform branch; the scale loads go
      // out behind the operand loads, so the in-order count of a later wait for the operands is the same on both paths.
      if (F16 && ld.k0 - ld.it.kbeg < NSETS * SBK) {
        load_scales<G, A_KMAJOR, TBM>(P(ld.it.pj).scale_a, soa, sa_);
        load_scales<G, B_KMAJOR, TBN>(P(ld.it.pj).scale_b, sob, sb_);
      }
      const int w_before = ld.w;
      advance(ld);
      if (ld.w != w_before) {  // uniform, no vector memory access inside
        if (GROUPED) {
          opA = P(ld.it.pj).A; opB = P(ld.it.pj).B;
          op_lda = P(ld.it.pj).lda; op_ldb = P(ld.it.pj).ldb;
        }
        item_offsets<G, A_KMAJOR, TBM>(op_lda, P(ld.it.pj).M, ld.it.bm0, pt, voa);
        item_offsets<G, B_KMAJOR, TBN>(op_ldb, P(ld.it.pj).N, ld.it.bn0, pt, vob);
        if (F16) {
          scale_offsets<G, A_KMAJOR, TBM>(P(ld.it.pj).M, ld.it.bm0, pt, soa, P(ld.it.pj).scale_a_stride);
          scale_offsets<G, B_KMAJOR, TBN>(P(ld.it.pj).N, ld.it.bn0, pt, sob, P(ld.it.pj).scale_b_stride);
        }
      }
    };
    auto fetch = [&](float4 (&a)[NVA], float4 (&b)[NVB], int &kskip, float (&sa_)[MAX_NV], float (&sb_)[MAX_NV]) __attribute__((always_inline)) {
      const int klim = ld.it.kend - ld.k0;
      const int ks = klim >= SBK ? ld.k0 : ld.it.kend - SBK;  // the last stage of an item may start early
      kskip = ld.k0 - ks;
      load_raw(opA + (A_KMAJOR ? (size_t)ks * op_lda : (size_t)ks), voa, a);
      load_raw(opB + (B_KMAJOR ? (size_t)ks * op_ldb : (size_t)ks), vob, b);
      fetch_tail(sa_, sb_);
    };
    // convert + store the stage under `st` into LDS buffer `buf`, then refill the registers two stages ahead
    auto produce = [&](float4 (&a)[NVA], float4 (&b)[NVB], int &kskip, float (&sa_)[MAX_NV], float (&sb_)[MAX_NV], int buf) __attribute__((always_inline)) {
      unsigned short *sa = smem + buf * STAGE, *sb = sa + G::NPLANES * PLANE_A;
      mask_tail<G, A_KMAJOR, TBM>(pt, a, kskip);
      mask_tail<G, B_KMAJOR, TBN>(pt, b, kskip);
      if (has_colsum) {  // kernel-uniform; the weights are zero where this workgroup has nothing to add
#pragma unroll
        for (int i = 0; i < NVA; ++i) {  // scalar fmas, kept apart: a v_pk_fma_f32 beside the consumer's MFMAs stalls the pipe
          csum.x = fmaf(cs_w[i], a[i].x, csum.x); asm volatile("" : "+v"(csum.x));
          csum.y = fmaf(cs_w[i], a[i].y, csum.y); asm volatile("" : "+v"(csum.y));
          csum.z = fmaf(cs_w[i], a[i].z, csum.z); asm volatile("" : "+v"(csum.z));
          csum.w = fmaf(cs_w[i], a[i].w, csum.w); asm volatile("" : "+v"(csum.w));
        }
      }
      if (F16) {
        // f16x2: a register is REFILLED (stage + NSETS) right behind its conversion, two float4 at a time, instead of all
        // loads behind all conversions: the load unit takes 16 cycles per 1 KB instruction and four wavefronts feed it, so
        // 12 loads issued back to back held a producer for 600 cycles with nothing to do (profiles/r03/r03_gemm_stage_trace.txt);
        // spread over the conversion they go out while the wavefront computes.  The registers keep their places (a float4
        // is loaded into the slot that was just converted), the in-order load count is the same on every path.
        const int klim = ld.it.kend - ld.k0;
        const int ks_next = klim >= SBK ? ld.k0 : ld.it.kend - SBK;  // the last stage of an item may start early
        const float *na = opA + (A_KMAJOR ? (size_t)ks_next * op_lda : (size_t)ks_next);
        const float *nb = opB + (B_KMAJOR ? (size_t)ks_next * op_ldb : (size_t)ks_next);
        constexpr int GRP = 2;
#pragma unroll
        for (int i0 = 0; i0 < NVA; i0 += GRP) {
#pragma unroll
          for (int i = i0; i < i0 + GRP && i < NVA; ++i) store_split_f16_one<G, A_KMAJOR, TBM>(sa, pt, a[i], sa_, i);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = i0; i < i0 + GRP && i < NVA; ++i)
            a[i] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(na) + voa[i]);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i0 = 0; i0 < NVB; i0 += GRP) {
#pragma unroll
          for (int i = i0; i < i0 + GRP && i < NVB; ++i) store_split_f16_one<G, B_KMAJOR, TBN>(sb, pt, b[i], sb_, i);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = i0; i < i0 + GRP && i < NVB; ++i)
            b[i] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(nb) + vob[i]);
          __builtin_amdgcn_sched_barrier(0);
        }
        kskip = ld.k0 - ks_next;
      } else {
        store_split<G, A_KMAJOR, TBM>(sa, pt, a);
        store_split<G, B_KMAJOR, TBN>(sb, pt, b);
      }
      if (has_colsum && !st.end && st.k0 + SBK >= st.it.kend) {  // last stage of its item: publish (LDS only)
        if (colsum_on(st.it)) {
          // Three rotating areas: an item's sums are published one stage before the consumers finish the item and read
          // by them after the barrier of its last stage; with items of a single stage the writer of item I + 2 may
          // already run while the reader of item I is still in its epilogue, the writer of item I + 3 may not.
          reinterpret_cast<float4 *>(cs_area + cs_parity * 4 * TBM)[pt] = csum;  // [k group = pt / 64][row quad]
          cs_parity = cs_parity == COLSUM_AREAS - 1 ? 0 : cs_parity + 1;
        }
        csum = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      {
        const int w_before = st.w;
        const bool was_end = st.end;
        advance(st);
        if (st.w != w_before) colsum_weights(st.it);
        if (st.end && !was_end) {  // past the end the last stage is re-fetched
#pragma unroll
          for (int i = 0; i < NVA; ++i) cs_w[i] = 0.f;
        }
      }
      // the refill must not be scheduled above the conversion: old and new contents of the registers would overlap,
      // the set could not stay in place across the loop and the copies (each waiting for its load) would drain the
      // prefetch queue every iteration
      __builtin_amdgcn_sched_barrier(0);
      if (F16) fetch_tail(sa_, sb_);
      else fetch(a, b, kskip, sa_, sb_);
    };
    // (the scheduling fences keep the ISSUE ORDER of the prologue loads: the scheduler would otherwise sink the later
    // fetches below the refill to shorten live ranges, and since vmcnt counts in order every later wait for an older
    // register set would have to drain the newer ones as well)
#pragma unroll
    for (int u = 0; u < NSETS; ++u) {
      fetch(ra[u], rb[u], rskip[u], rsa[u], rsb[u]);
      __builtin_amdgcn_sched_barrier(0);
    }
    produce(ra[0], rb[0], rskip[0], rsa[0], rsb[0], 0);  // stage 0 -> buffer 0, set 0 <- stage NSETS
    __syncthreads();
    for (int g = 0; g < padded_stages; g += NSETS) {  // no exit in the middle: the register sets keep their roles
#pragma unroll
      for (int u = 1; u <= NSETS; ++u) {
        produce(ra[u % NSETS], rb[u % NSETS], rskip[u % NSETS], rsa[u % NSETS], rsb[u % NSETS], u & 1);  // stage g + u -> buffer (g + u) & 1
        __syncthreads();
      }
    }
  } else {
    // ================================================================ consumers
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[TI][2];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();
    Cursor cc = {work.begin, 0, item_at(work.begin), false};
    cc.k0 = cc.it.kbeg;
    int cs_parity = 0;  // which of the published column-sum areas belongs to the current item

    __syncthreads();  // stage 0 is in buffer 0
    for (int g = 0; g < padded_stages; ++g) {
      if (g >= total_stages) {  // padding stage: the producers' loop runs in groups of NSETS stages
        __syncthreads();
        continue;
      }
      const unsigned short *sa = smem + (g & 1) * STAGE, *sb = sa + G::NPLANES * PLANE_A;
      bf16x8 fa[TI][3], fb[2][3];
      int ks = 0;  // 16-k step of the stage
      auto read_a = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TI; ++i) fa[i][t] = read_frag<G, A_KMAJOR, TBM>(sa, wm * (32 * TI) + 32 * i, lane, t, ks);
      };
      auto read_b = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j][t] = read_frag<G, B_KMAJOR, TBN>(sb, wn * 64 + 32 * j, lane, t, ks);
      };
      auto mul = [&](int ta, int tb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = F16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i][ta]),
                                                                     __builtin_bit_cast(f16x8, fb[j][tb]), acc[i][j], 0, 0, 0)
                            : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ta], fb[j][tb], acc[i][j], 0, 0, 0);
      };
      // smallest products first; fragments are read in the order the products need them
      if (NPROD == 3) {
#pragma unroll
        for (int kk = 0; kk < SBK / 16; ++kk) {
          ks = kk;
          read_a(1); read_b(0); mul(1, 0);
          read_a(0); read_b(1); mul(0, 1);
          mul(0, 0);
        }
      } else if (NPROD == 9) {
        read_a(2); read_b(2); mul(2, 2);
        read_b(1); mul(2, 1);
        read_a(1); mul(1, 2);
        read_b(0); mul(2, 0);
        read_a(0); mul(0, 2);
      } else {
        read_a(2); read_b(0); mul(2, 0);
        read_a(0); read_b(2); mul(0, 2);
        read_a(1); read_b(1);
      }
      if (NPROD != 3) { mul(1, 1); mul(1, 0); mul(0, 1); mul(0, 0); }
      __syncthreads();  // buffer g & 1 is released, buffer (g + 1) & 1 holds stage g + 1
      if (cc.k0 + SBK >= cc.it.kend) {  // that was the item's last stage
        const GemmParams &p = P(cc.it.pj);
        const uint32_t thr = dropout_threshold(p.dropout_p);
        const float keep_scale = 1.f / (1.f - p.dropout_p);
        const int cs_share = cs_share_of(cc.it);
        float *C = p.C + (partial ? (size_t)cc.it.z * p.slab : 0);
        const int ldc = partial ? p.N : p.ldc;
        const int row0 = cc.it.bm0 + wm * (32 * TI), col0 = cc.it.bn0 + wn * 64;
        if (F16) {  // back from the scaled operands: acc / (scale_a[row] scale_b[col]), exact (powers of two)
          const int l31 = lane & 31, lh = lane >> 5;
          float ib[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) ib[j] = inverse_scale(p.scale_b[min(col0 + j * 32 + l31, p.N - 1) * p.scale_b_stride]);
#pragma unroll
          for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float ia = inverse_scale(p.scale_a[min(row0 + i * 32 + 8 * g + 4 * lh + e, p.M - 1) * p.scale_a_stride]);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j][g * 4 + e] = acc[i][j][g * 4 + e] * ia * ib[j];
              }
        }
        if (p.vec_epilogue) tile_epilogue_vec<TI, true, EPI, NPROD == 3 && EPI == EPI_NODROP>(p, acc, C, ldc, partial, row0, col0, lane, thr, keep_scale, scratch + wave * 2048);
        else tile_epilogue_vec<TI, false, EPI>(p, acc, C, ldc, partial, row0, col0, lane, thr, keep_scale, scratch + wave * 2048);
        zero_acc();
        if (colsum_on(cc.it)) {  // the producers published this item's sums before the barrier above: one row per lane
          const float *q = cs_area + cs_parity * 4 * TBM;
          const int row = cc.it.bm0 + tid;
          if (row < p.M) {
            const float tot = q[tid] + q[TBM + tid] + q[2 * TBM + tid] + q[3 * TBM + tid];
            if (partial) p.colsum[((size_t)cc.it.z * cs_share + colsum_first(cc.it)) * p.M + row] = tot;
            else p.colsum[row] += tot;
          }
          cs_parity = cs_parity == COLSUM_AREAS - 1 ? 0 : cs_parity + 1;
        }
      }
      advance(cc);
    }
  }
}

constexpr int TI1_COST_X10 = 4;   // cost of a 64-row tile in tenths of a 256-row tile (so 2 * TI2_COST_NUM / 2 * TI2_COST_DEN are the others)
constexpr int TI2_COST_NUM = 3, TI2_COST_DEN = 5;  // cost of a 128-row tile / a 256-row tile: 0.55 - 0.64 measured (profiles/r03/r03_tile_height.txt)
template <bool AK, bool BKM, int NPROD, int EPI, int TI>
int launch_ti(const GemmParams &p, int splits, hipStream_t st) {
  using G = Geo<NPROD, TI>;
  const int work = ((p.M + G::TBM - 1) / G::TBM) * ((p.N + TBN - 1) / TBN) * splits;
  auto kern = gemm_bf16x3_mfma_kernel<AK, BKM, NPROD, EPI, TI>;
  PT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)G::LDS_BYTES));  // idempotent, host-only: no state kept between calls
  const int slots = persistent_grid(p.reserved_cus);  // one workgroup per CU
  const int grid = work < slots ? work : slots;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), G::LDS_BYTES, st, p);
  return pt_check_launch();
}
template <bool AK, bool BKM, int NPROD, int EPI>
int launch(const GemmParams &p, int splits, hipStream_t st) {
  // 128-row tiles where they finish sooner: rounds of tiles over the CUs x the cost of a tile (K-contiguous A only).
  if (!AK && NPROD != 9 && p.M > 128) {
    const int slots = persistent_grid(p.reserved_cus), nt = (p.N + TBN - 1) / TBN * splits;
    const int rounds4 = (((p.M + 255) / 256) * nt + slots - 1) / slots, rounds2 = (((p.M + 127) / 128) * nt + slots - 1) / slots;
    // 64-row tiles (f16x2 only; round 6) where even 128-row tiles leave CUs without one: the per-GPU share of a strongly
    // scaled batch (2048 / 4096 tokens), N = 512.  A lone tile's time is its stage count x the stage time, and that falls with
    // the tile height - 0.72 us per 32-k stage at 128 rows, 0.50 at 64 (profiles/r06/r06_longk_*.txt): wo forward / dX of wo
    // 21.7 -> 16.0 us at 2048 tokens, 24.1 -> 17.7 at 4096.  Cost 0.4 of a 256-row tile: never chosen when it needs a second round.
    if constexpr (NPROD == 3) {
      const int rounds1 = (((p.M + 63) / 64) * nt + slots - 1) / slots;
      if (TI1_COST_X10 * rounds1 < 2 * TI2_COST_NUM * rounds2 && TI1_COST_X10 * rounds1 < 2 * TI2_COST_DEN * rounds4)
        return launch_ti<false, BKM, NPROD, EPI, 1>(p, splits, st);
    }
#if defined(PT_FORCE_TI)
    if (PT_FORCE_TI == 2) return launch_ti<false, BKM, NPROD, EPI, 2>(p, splits, st);
#else
    if (TI2_COST_NUM * rounds2 < TI2_COST_DEN * rounds4) return launch_ti<false, BKM, NPROD, EPI, 2>(p, splits, st);
#endif
  }
  return launch_ti<AK, BKM, NPROD, EPI, 4>(p, splits, st);
}

// a group (k-major x k-major, 256-row tiles, slabs): one persistent launch over the members' concatenated work items
template <int NPROD>
int launch_group(const GemmGroup &g, hipStream_t st) {
  using G = Geo<NPROD, 4>;
  auto kern = gemm_bf16x3_mfma_kernel<true, true, NPROD, EPI_PLAIN, 4, true>;
  PT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)G::LDS_BYTES));
  const int slots = persistent_grid(g.p[0].reserved_cus), work = g.first[g.n];
  hipLaunchKernelGGL(kern, dim3(work < slots ? work : slots), dim3(NTHREADS), G::LDS_BYTES, st, g);
  return pt_check_launch();
}

template <int NPROD, int EPI>
int launch_layout(const GemmParams &p, bool ak, bool bk, int splits, hipStream_t st) {
  if (!ak && !bk) return launch<false, false, NPROD, EPI>(p, splits, st);
  if (!ak && bk) return launch<false, true, NPROD, EPI>(p, splits, st);
  if (ak && !bk) return launch<true, false, NPROD, EPI>(p, splits, st);
  return launch<true, true, NPROD, EPI>(p, splits, st);
}

}  // namespace

}  // namespace ptgemm
