// ptamd_gemm_hp: fp32-grade GEMM on the f16 matrix pipe from PRE-SPLIT operands (hp_format.h), and the kernels that write
// that format from fp32 tensors.
//
// Same role, epilogues and dropout masks as ptamd_gemm (every torch.nn.Linear of the reference encoder and its backward
// dX products: Attention.py:38-41,49,69; Sublayers.py:28-34; encoder_only.py:18,39-41) in the PTAMD_GEMM_F16X2
// arithmetic (include/ptamd.h):  C = A B^T  with both operands K-contiguous, evaluated as the three products
// hi hi' + hi lo' + lo hi' of v_mfma_f32_32x32x16_f16 with f32 accumulation, the two row scales taken out of the
// accumulators (exactly: powers of two) in front of the epilogue.
//
// What is different from gemm_split_kernel.h (which splits fp32 operands while it stages them): nothing is converted
// here.  Operands arrive as two f16 planes in 32 x 16 blocks that are byte-for-byte the LDS image of an MFMA operand,
// so a stage is filled by `global_load_lds_dwordx4` (LDS-DMA: no VGPRs, no VALU, no ds_write - the conversion chain, the
// LDS store traffic and half of the wavefronts of the old kernel existed only for that) and all eight wavefronts of the
// workgroup issue MFMAs.
//
// Structure: one persistent workgroup per CU (512 threads = 8 wavefronts, two per SIMD), 256 x 128 output tile, wavefront
// (wm, wn) owns 64 x 64 = 2 x 2 MFMA tiles (64 accumulator VGPRs).  A stage is 32 k: 32 KiB of A + 16 KiB of B, 48
// wave-level 1-KiB DMA pieces, six per wavefront; two stage buffers; the pieces of stage q + 1 are issued right after the
// barrier that opens stage q and land while the 24 MFMAs per wavefront of stage q run (one `s_waitcnt vmcnt(0)` +
// `s_barrier` per stage).  Per 16 k a wavefront reads 8 fragments (ds_read_b128, conflict-free by the chunk swizzle) for
// 12 MFMAs.  Work items (tile x K split) are walked in contiguous XCD-aware ranges like the other kernels; the stage
// stream runs across items, so the first stage of the next tile is in flight under the epilogue of this one.
// Epilogue: the shared tile_epilogue_vec (bias -> ReLU -> dropout -> residual / gate -> tanh -> accumulate, float4 rows
// through a per-wavefront LDS transpose).
#include <stdlib.h>

#include "gemm_common.h"
#include "hp_format.h"
#include "kv_format.h"
#include "split_bf16.h"

// v_writelane_b32 through the LLVM intrinsic (no clang builtin here; as inline asm the compiler would not see its hazards)
extern "C" __device__ int pt_llvm_amdgcn_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane");

namespace pthp {
namespace {

using ptgemm::GemmParams;
using ptgemm::f32x16;
using ptsplit::f16x8;

constexpr int HBN = 128;  // tile columns: two wavefront columns of 64

struct HpParams {
  GemmParams g;            // shapes, C, epilogue operands, split-K bookkeeping (A / B / lda / ldb unused)
  const char *a_planes, *b_planes;
  const float *a_scale, *b_scale;
  int kb16;                // 16-column blocks per block row (= Kp / 16)
  int a_rb_last, b_rb_last;  // last valid block row of each operand (tile rows beyond are clamped, never stored)
  // the QKV product (Attention.py:49): the columns from kv_col0 on - K and V, head by head, 64 columns each - leave the epilogue
  // as pre-split planes (kv_format.h) instead of fp32; columns in front (Q) are stored as usual
  char *kv_planes;
  float *kv_inv;
  int kv_col0, kv_heads, kv_nt;
};

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
__device__ __forceinline__ void dma16(const char *g, char *l) {  // 64 lanes x 16 B -> 1 KiB of LDS at l (wave-uniform) + 16 lane
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

struct Item {
  int bm0, bn0, z, kbeg, kend;
};
struct Cursor {
  int w, k0;
  Item it;
};

// Geometry of a variant: WM wavefront rows x 2 wavefront columns; a wavefront owns (32 TI) x 64 of the (32 TI WM) x 128
// tile; a stage is 16 KB16 k.  ILV: the DMA pieces of the next stage are issued between the MFMA groups of this one
// instead of in a burst behind the barrier (an LDS-DMA issue occupies the wavefront's instruction stream for ~100 cycles).
template <int WM_, int TI_, int KB_, bool ILV_>
struct HpGeom {
  static constexpr int WM = WM_, TI = TI_, KB = KB_;
  static constexpr bool ILV = ILV_;
  static constexpr int NW = 2 * WM, THREADS = 64 * NW, TILE_M = 32 * TI * WM, BK = 16 * KB;
  static constexpr int A_BLOCKS = TILE_M / 32, B_BLOCKS = HBN / 32;
  static constexpr int RB_BYTES = KB * 2048;                       // one block row of a stage: KB x (hi, lo) x 1 KiB
  static constexpr int A_STAGE = A_BLOCKS * RB_BYTES, STAGE_BYTES = (A_BLOCKS + B_BLOCKS) * RB_BYTES;
  static constexpr int PIECES = STAGE_BYTES / 1024, PER_WAVE = PIECES / NW;
  static constexpr int SCRATCH_BYTES = NW * 2048 * 4;
  static constexpr size_t LDS = (size_t)2 * STAGE_BYTES + SCRATCH_BYTES;
  static constexpr int WG_PER_CU = LDS <= 80 * 1024 ? 2 : 1;
  static_assert(PIECES % NW == 0, "pieces must divide evenly over the wavefronts");
  static_assert(LDS <= 160 * 1024, "LDS budget of a CU");
};

template <typename G, int EPI>
__global__ __launch_bounds__(G::THREADS, G::WG_PER_CU * G::NW / 4) void gemm_hp_kernel(const HpParams p) {
  constexpr int TI = G::TI, KB = G::KB, BK = G::BK;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *const scratch = reinterpret_cast<float *>(smem + 2 * G::STAGE_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  // the wavefront index as a SCALAR: everything derived from it (which DMA piece, which operand, LDS destinations) is then
  // computed on the scalar unit and the operand base pointer is a scalar select - left as a vector value the compiler
  // re-loaded the selected pointer from the kernel-argument segment inside every DMA slot and waited vmcnt(0) for it,
  // draining the LDS-DMA queue in the middle of the stage
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const ptgemm::WorkRange work(p.g, G::TILE_M, HBN);
  if (work.begin >= work.end) return;
  const int Kp = p.kb16 * 16;
  auto item_at = [&](int logical) __attribute__((always_inline)) {
    Item it;
    work.decode(logical, it.bm0, it.bn0, it.z);
    it.kbeg = it.z * p.g.k_per_split;
    it.kend = min(Kp, it.kbeg + p.g.k_per_split);
    return it;
  };
  // next stage of the stream; false past the last one
  auto advance = [&](Cursor &c) __attribute__((always_inline)) {
    if (c.k0 + BK < c.it.kend) {
      c.k0 += BK;
      return true;
    }
    if (c.w + 1 < work.end) {
      c.it = item_at(++c.w);
      c.k0 = c.it.kbeg;
      return true;
    }
    return false;
  };
  // DMA piece i (0 .. PER_WAVE-1) of this wavefront for stage (item, k0) into stage buffer `buf`.  Piece q = wave +
  // NW i of the stage covers 1 KiB: stage offset q KiB = [block row][k block][plane], the same order as in memory.
  const int lane16 = lane * 16;
  auto issue_piece = [&](const Cursor &c, int buf, int i) __attribute__((always_inline)) {
    const int q = wave + G::NW * i;
    const int rbq = q / (2 * KB), rest = q - rbq * (2 * KB);          // block row of the stage, (k block, plane) inside it
    const bool is_b = rbq >= G::A_BLOCKS;
    const int rb = is_b ? min((c.it.bn0 >> 5) + rbq - G::A_BLOCKS, p.b_rb_last) : min((c.it.bm0 >> 5) + rbq, p.a_rb_last);
    const char *g = (is_b ? p.b_planes : p.a_planes) + block_offset(rb, c.k0 >> 4, 0, p.kb16) + rest * 1024 + lane16;
    dma16(g, smem + buf * G::STAGE_BYTES + q * 1024);
  };

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
  const uint32_t thr = dropout_threshold(p.g.dropout_p);
  const float keep_scale = 1.f / (1.f - p.g.dropout_p);
  const bool partial = p.g.slab != 0;
  const int frag_off = chunk_index(lane & 31, lane >> 5) * 16;  // this lane's 16-byte chunk inside every block

  Cursor ld = {work.begin, 0, item_at(work.begin)};
  ld.k0 = ld.it.kbeg;
  Cursor cc = ld;
#pragma unroll
  for (int i = 0; i < G::PER_WAVE; ++i) issue_piece(ld, 0, i);
  bool more_loads = advance(ld);
  int buf = 0;
  for (;;) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's pieces of the stage have landed ...
    __syncthreads();                                   // ... and everybody's; everybody is done with the other buffer
    if (!G::ILV && more_loads) {
#pragma unroll
      for (int i = 0; i < G::PER_WAVE; ++i) issue_piece(ld, buf ^ 1, i);
    }
    const char *sa = smem + buf * G::STAGE_BYTES + (TI * wm) * G::RB_BYTES + frag_off;
    const char *sb = smem + buf * G::STAGE_BYTES + G::A_STAGE + (2 * wn) * G::RB_BYTES + frag_off;
    int piece = 0;
    auto dma_slot = [&](int n) __attribute__((always_inline)) {  // ILV: n pieces of the next stage here
      if (G::ILV) {
        __builtin_amdgcn_sched_barrier(0);
        if (more_loads) {
#pragma unroll
          for (int u = 0; u < n; ++u)
            if (piece + u < G::PER_WAVE) issue_piece(ld, buf ^ 1, piece + u);
        }
        piece += n;
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    constexpr int SLOTS = 3 * KB, PER_SLOT = (G::PER_WAVE + SLOTS - 1) / SLOTS;
    f16x8 fa[TI][2], fb[2][2];  // [tile][plane]
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int i = 0; i < TI; ++i) fa[i][t] = *reinterpret_cast<const f16x8 *>(sa + i * G::RB_BYTES + (kb * 2 + t) * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j][t] = *reinterpret_cast<const f16x8 *>(sb + j * G::RB_BYTES + (kb * 2 + t) * 1024);
      }
      // smallest products first
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
      dma_slot(PER_SLOT);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
      dma_slot(PER_SLOT);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
      dma_slot(PER_SLOT);
    }
    if (more_loads) more_loads = advance(ld);
    if (cc.k0 + BK >= cc.it.kend) {  // that was the item's last stage (uniform)
      float *C = p.g.C + (partial ? (size_t)cc.it.z * p.g.slab : 0);
      const int ldc = partial ? p.g.N : p.g.ldc;
      const int row0 = cc.it.bm0 + wm * 32 * TI, col0 = cc.it.bn0 + wn * 64;
      {  // back from the scaled operands: acc / (scale_a[row] scale_b[col]), exact (powers of two)
        const int l31 = lane & 31, lh = lane >> 5;
        float ib[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) ib[j] = inverse_of_scale(p.b_scale[min(col0 + j * 32 + l31, p.g.N - 1)]);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float ia = inverse_of_scale(p.a_scale[min(row0 + i * 32 + 8 * g + 4 * lh + e, p.g.M - 1)]);
#pragma unroll
              for (int j = 0; j < 2; ++j) acc[i][j][g * 4 + e] = acc[i][j][g * 4 + e] * ia * ib[j];
            }
      }
      if (p.g.vec_epilogue)
        ptgemm::tile_epilogue_vec<TI, true, EPI>(p.g, acc, C, ldc, partial, row0, col0, lane, thr, keep_scale, scratch + wave * 2048);
      else
        ptgemm::tile_epilogue_vec<TI, false, EPI>(p.g, acc, C, ldc, partial, row0, col0, lane, thr, keep_scale, scratch + wave * 2048);
      zero_acc();
    }
    if (!advance(cc)) break;
    buf ^= 1;
  }
}

template <typename G, int EPI>
int launch_hp_g(const HpParams &p, int splits, hipStream_t st) {
  const int work = ((p.g.M + G::TILE_M - 1) / G::TILE_M) * ((p.g.N + HBN - 1) / HBN) * splits;
  auto kern = gemm_hp_kernel<G, EPI>;
  PT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS));
  const int slots = ptgemm::persistent_grid(p.g.reserved_cus) * G::WG_PER_CU;
  hipLaunchKernelGGL(kern, dim3(work < slots ? work : slots), dim3(G::THREADS), G::LDS, st, p);
  return pt_check_launch();
}

// The geometry in use: 8 wavefronts x (64 x 64) = 256 x 128 tile, 32 k per stage, one workgroup per CU, DMA pieces between
// the MFMA groups.  Measured alternatives (profiles/r02/r02_hp_gemm_ablations.txt): the DMA burst behind the barrier
// (HpGeom<4, 2, 2, false>) is 5-8 % slower; two workgroups of 4 wavefronts x (128 x 64) per CU (HpGeom<2, 4, 1, true>) 20-40 %
// slower and at the 256-VGPR limit; 128 x 128 tiles (HpGeom<2, 2, 2, true>) 30-50 % slower.
typedef HpGeom<4, 2, 2, true> Geom;

template <int EPI>
int launch_hp3(const HpParams &p, int splits, hipStream_t st);
// the three-stage kernel (below) wherever the float4 epilogue applies; PTAMD_HP_STAGES=2 in the environment (read at every
// call) selects the two-buffer kernel for A/B measurements
template <int EPI>
int launch_hp(const HpParams &p, int splits, hipStream_t st) {
  const char *e = getenv("PTAMD_HP_STAGES");
  // (the no-dropout epilogue instantiation of the three-stage kernel spills 45 registers - its epilogue has no generator
  // loop to keep the two 32-row blocks apart - so those launches take the full instantiation, whose dropout branch is a
  // run-time no-op at p = 0: same results, no spills)
  if (p.g.vec_epilogue && !(e && e[0] == '2')) return launch_hp3<EPI == ptgemm::EPI_NODROP ? ptgemm::EPI_FULL : EPI>(p, splits, st);
  if (p.g.gate_mask_out) return PTAMD_ERR_BAD_SHAPE;   // only the three-stage kernel's epilogue writes the 1-bit gate
  return launch_hp_g<Geom, EPI>(p, splits, st);
}

// ---------------------------------------------------------------------------------------------- three-stage variant (round 4)
// Same tile (256 x 128, eight wavefronts of 64 x 64), same stages of 32 k, same epilogue arithmetic and masks as
// gemm_hp_kernel above, with the stage stream restructured around what the ISA of that kernel showed:
//   * THREE stage buffers: the pieces of stage s + 2 are issued while stage s computes, and the wait at the top of a
//     stage is a COUNTED `s_waitcnt vmcnt(6)` - this wavefront's six pieces of stage s + 1 stay in flight across the
//     barrier (a raw s_barrier: __syncthreads() would drain the LDS-DMA queue).  In the two-buffer kernel the last piece
//     of the next stage is issued near the end of a stage and `vmcnt(0)` at the top waits out its whole latency - every
//     32 k.  LDS: 3 x 48 KiB of stages + 16 KiB of epilogue scratch (2 KiB per wavefront: the accumulators go out eight
//     rows at a time) = 160 KiB;
//   * a wavefront's six pieces are the same (operand, block row, k block, plane) in every stage: four of A, two of B.
//     Their addresses are advanced on the scalar unit from values computed once per tile; the old kernel re-loaded the
//     selected operand pointer from the kernel-argument segment in every DMA slot and waited lgkmcnt(0) for it - which
//     also waited for the fragment reads in flight;
//   * the sixteen fragment reads of a stage are issued in one burst behind the barrier (inline asm, counted lgkmcnt): the
//     first twelve MFMAs start when the first eight have returned.
// Two geometries: WM = 4 - eight wavefronts, 256 x 128 tile, three stage buffers, one workgroup per CU (160 KiB); WM = 2 - four
// wavefronts, 128 x 128 tile, two stage buffers, TWO workgroups per CU (72 KiB each): the tile epilogue of one workgroup (LDS
// transposes, dropout generator, gate / residual reads, the tile's stores) runs beside the main loop of the other instead
// of leaving the matrix pipe idle - for the K = 512 products, whose epilogue is a third of a tile's time.
template <int WM_>
struct Hp3G {
  static constexpr int WM = WM_, NW = 2 * WM, THREADS = 64 * NW, TILE_M = 64 * WM, BK = 32, NSTAGE = WM == 4 ? 3 : 2;
  static constexpr int A_BLOCKS = 2 * WM, B_BLOCKS = 4, PER_WAVE = (A_BLOCKS + B_BLOCKS) * 4 / NW;   // 6 or 8 pieces of 1 KiB
  static constexpr int RB_BYTES = 4096, A_STAGE = A_BLOCKS * RB_BYTES, STAGE_BYTES = (A_BLOCKS + B_BLOCKS) * RB_BYTES;
  static constexpr int SCRATCH_BYTES = NW * 2048;
  static constexpr size_t LDS = (size_t)NSTAGE * STAGE_BYTES + SCRATCH_BYTES;
  static constexpr int WG_PER_CU = WM == 4 ? 1 : 2;
  static_assert(LDS * WG_PER_CU <= 160 * 1024, "the LDS of a CU");
};
typedef Hp3G<4> Hp3;

// the epilogue of one wavefront's 64 x 64 block (TI = 2), eight rows at a time through `scratch` (512 floats): bias / ReLU /
// dropout in the MFMA layout, residual / gate / accumulate operands and the stores as float4 rows.  Arithmetic, order of
// operations and dropout masks are those of ptgemm::tile_epilogue_vec (gemm_common.h).
// eight f32 of one row -> the two f16x8 chunks (hi, lo) of x * s, with the arithmetic of the attention kernels' own staging
__device__ __forceinline__ void kv_split8(const float4 &a, const float4 &b, float s, uint4 &hi, uint4 &lo) {
  uint2 h0, l0, h1, l1;
  ptsplit::split_quad_f16(a.x, a.y, a.z, a.w, s, s, s, s, h0, l0);
  ptsplit::split_quad_f16(b.x, b.y, b.z, b.w, s, s, s, s, h1, l1);
  hi = make_uint4(h0.x, h0.y, h1.x, h1.y);
  lo = make_uint4(l0.x, l0.y, l1.x, l1.y);
}
// maximum over the 32 lanes of a wavefront half (in every lane of it): four DPP steps inside the rows of 16, one exchange
__device__ __forceinline__ uint32_t half_umax(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));  // row_half_mirror
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));  // row_mirror
  v = max(v, (uint32_t)__shfl_xor((int)v, 16, 64));
  return v;
}

template <int EPI, bool KVP = false>
__device__ __forceinline__ void hp3_epilogue(const GemmParams &p, const f32x16 (&acc)[2][2], float *C, int ldc, bool partial,
                                             int row0, int col0, int lane, uint32_t thr, float keep_scale, float *scratch,
                                             const HpParams *hp = nullptr) {
  using ptgemm::EPI_FULL;
  using ptgemm::EPI_PLAIN;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int l31 = lane & 31, lh = lane >> 5;
  uint32_t keep[2] = {0xffffffffu, 0xffffffffu};
  if (EPI == EPI_FULL && !partial && p.dropout_p > 0.f) {
    keep[0] = keep[1] = 0u;
#pragma unroll 4
    for (int idx = 0; idx < 8; ++idx) {  // one call = the lane's 8 rows of two register groups (common.h)
      const int i = idx >> 2, j = (idx >> 1) & 1, gp = idx & 1;
      const int row = row0 + i * 32 + 16 * gp + 4 * lh, col = col0 + j * 32 + l31;
      const uint4 rnd = pt_rand4(p.seed, drop_call_index(row, col, p.N), p.stream_id);
      const uint32_t t16 = thr >> 16;
      uint32_t bits = 0;
#pragma unroll
      for (int f = 0; f < 8; ++f)
        bits |= (drop_field_value(rnd, f) >= t16 ? 1u : 0u) << (j * 16 + (2 * gp + (f >> 2)) * 4 + (f & 3));
      keep[0] |= i == 0 ? bits : 0u;
      keep[1] |= i == 1 ? bits : 0u;
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    // (one 32-row block at a time: without the fence the compiler requests the residual / gate operands of BOTH blocks up
    // front - 64 registers - and the no-dropout instantiation, which has no generator loop between them, spills 45)
    __builtin_amdgcn_sched_barrier(0);
    f32x4 pre4[8];
    const bool want_res = EPI != EPI_PLAIN && !partial && p.residual != nullptr;
    const bool want_old = EPI != EPI_PLAIN && !partial && (p.flags & PTAMD_EPI_ACCUM) && !want_res;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int f = lane + 64 * t, rr = f >> 4, c4 = (f & 15) * 4;
      const int row = min(row0 + i * 32 + rr, p.M - 1), col = min(col0 + c4, p.N - 4);
      const float *src = want_res ? p.residual + (size_t)row * p.ldr + col : C + (size_t)row * ldc + col;
      pre4[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (want_res || want_old) pre4[t] = *reinterpret_cast<const f32x4 *>(src);
    }
    float bias[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = col0 + j * 32 + l31;
      bias[j] = (EPI != EPI_PLAIN && !partial && p.bias && col < p.N) ? p.bias[col] : 0.f;
    }
    int mask_lo[2] = {0, 0}, mask_hi[2] = {0, 0};
    const bool gated = EPI != EPI_PLAIN && !partial && p.gate_mask != nullptr;   // (uniform)
    uint32_t gmk[2] = {0u, 0u};
    if (gated) {
      gmk[0] = ptgemm::load_gate_masks(p, (row0 >> 5) + i, col0 >> 5, lane);
      gmk[1] = ptgemm::load_gate_masks(p, (row0 >> 5) + i, (col0 >> 5) + 1, lane);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[i][j][g * 4 + e];
          if (EPI != EPI_PLAIN && !partial) {
            v += bias[j];
            if (p.flags & PTAMD_EPI_RELU) v = fmaxf(v, 0.f);
            if (EPI == EPI_FULL && p.dropout_p > 0.f) v = (keep[i] >> (j * 16 + g * 4 + e)) & 1u ? v * keep_scale : 0.f;
            if (gated) v = ptgemm::gate_keep(gmk[j], g * 4 + e) ? v * p.gate_scale : 0.f;
            if (p.gate_mask_out) {   // (uniform) the decisions "result > 0" of this register: lane g * 4 + e of the block's pair
              const uint64_t m = __builtin_amdgcn_ballot_w64(v > 0.f);
              mask_lo[j] = pt_llvm_amdgcn_writelane((int)(uint32_t)m, g * 4 + e, mask_lo[j]);
              mask_hi[j] = pt_llvm_amdgcn_writelane((int)(uint32_t)(m >> 32), g * 4 + e, mask_hi[j]);
            }
          }
          scratch[(4 * lh + e) * 64 + j * 32 + l31] = v;
        }
      if (KVP && col0 >= hp->kv_col0) {
        // K / V columns (one head per wavefront): the eight rows go out as planes - lane (row lane >> 3 of the eight, chunk
        // lane & 7) holds 8 consecutive d; a group of four rows = a wavefront half, scaled by the power of two of ITS maximum
        const int rr = lane >> 3, ch = lane & 7;
        const float4 a = *reinterpret_cast<const float4 *>(scratch + rr * 64 + ch * 8);
        const float4 b = *reinterpret_cast<const float4 *>(scratch + rr * 64 + ch * 8 + 4);
        const uint32_t am = half_umax(max(max(max(__float_as_uint(a.x) & 0x7fffffffu, __float_as_uint(a.y) & 0x7fffffffu),
                                              max(__float_as_uint(a.z) & 0x7fffffffu, __float_as_uint(a.w) & 0x7fffffffu)),
                                          max(max(__float_as_uint(b.x) & 0x7fffffffu, __float_as_uint(b.y) & 0x7fffffffu),
                                              max(__float_as_uint(b.z) & 0x7fffffffu, __float_as_uint(b.w) & 0x7fffffffu))));
        const uint32_t sbits = pt_row_scale_bits(am);
        uint4 hi, lo;
        kv_split8(a, b, __uint_as_float(sbits), hi, lo);
        const int hc = (col0 - hp->kv_col0) >> 6;                       // (which, head): which = hc / heads
        const int gt = (row0 >> 5) + i, r = 8 * g + rr;
        if (row0 + i * 32 < p.M) {   // (M is a multiple of 32 on this path: whole tiles only)
          char *tile = hp->kv_planes + ((size_t)hc * hp->kv_nt + gt) * ptkv::TILE_BYTES;
          char *dst = tile + r * ptkv::ROW_BYTES + ptkv::chunk_pos(r, ch) * 16;
          *reinterpret_cast<uint4 *>(dst) = hi;
          *reinterpret_cast<uint4 *>(dst + ptkv::PLANE_BYTES) = lo;
          if ((lane & 31) == 0)
            hp->kv_inv[((size_t)hc * hp->kv_nt + gt) * 8 + ptkv::group_slot(2 * g + lh)] = __uint_as_float((254u << 23) - sbits);
        }
        continue;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int f = lane + 64 * u, rr = f >> 4, c4 = (f & 15) * 4;   // rr: 0..7 of this group of eight rows
        const int row = row0 + i * 32 + 8 * g + rr, col = col0 + c4;
        float4 v = *reinterpret_cast<const float4 *>(scratch + rr * 64 + c4);
#ifdef PT_ABLATE_NOSTORE   // ablation build: everything but the global store of the tile (the condition is never true)
        if (row < p.M && col < p.N && p.M < 0) {
#else
        if (row < p.M && col < p.N) {
#endif
          if (EPI != EPI_PLAIN && !partial) {
            // (pre4[t] of tile_epilogue_vec holds rows (lane >> 4) + 4 t of the 32-row block: t = 2 g + u here)
            const f32x4 r4 = pre4[2 * g + u];
            if (p.residual) {
              if (p.flags & PTAMD_EPI_GATE) {
                v.x = r4.x > 0.f ? v.x * p.gate_scale : 0.f; v.y = r4.y > 0.f ? v.y * p.gate_scale : 0.f;
                v.z = r4.z > 0.f ? v.z * p.gate_scale : 0.f; v.w = r4.w > 0.f ? v.w * p.gate_scale : 0.f;
              } else {
                v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
              }
            }
            if (p.flags & PTAMD_EPI_TANH) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
            if (p.flags & PTAMD_EPI_ACCUM) {
              const f32x4 o4 = want_old ? r4 : *reinterpret_cast<const f32x4 *>(C + (size_t)row * ldc + col);
              v.x += o4.x; v.y += o4.y; v.z += o4.z; v.w += o4.w;
            }
          }
          *reinterpret_cast<float4 *>(C + (size_t)row * ldc + col) = v;
        }
      }
    }
    if (EPI != EPI_PLAIN && !partial && p.gate_mask_out) {
      const int rb = __builtin_amdgcn_readfirstlane((row0 >> 5) + i);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cb = __builtin_amdgcn_readfirstlane((col0 >> 5) + j);
        if (lane < 16 && rb < p.mask_rb && cb < p.mask_cb)
          p.gate_mask_out[((size_t)cb * p.mask_rb + rb) * 16 + lane] = ((uint64_t)(uint32_t)mask_hi[j] << 32) | (uint32_t)mask_lo[j];
      }
    }
  }
}

#define PT_DS_READ_B128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

template <int EPI, int WM, bool KVP = false>
__global__ __launch_bounds__(Hp3G<WM>::THREADS, 2) void gemm_hp3_kernel(const HpParams p) {
  using G = Hp3G<WM>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *const scratch = reinterpret_cast<float *>(smem + G::NSTAGE * G::STAGE_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const ptgemm::WorkRange work(p.g, G::TILE_M, HBN);
  if (work.begin >= work.end) return;
  const int Kp = p.kb16 * 16;
  auto item_at = [&](int logical) __attribute__((always_inline)) {
    Item it;
    work.decode(logical, it.bm0, it.bn0, it.z);
    it.kbeg = it.z * p.g.k_per_split;
    it.kend = min(Kp, it.kbeg + p.g.k_per_split);
    return it;
  };
  auto advance = [&](Cursor &c) __attribute__((always_inline)) {
    if (c.k0 + G::BK < c.it.kend) {
      c.k0 += G::BK;
      return true;
    }
    if (c.w + 1 < work.end) {
      c.it = item_at(++c.w);
      c.k0 = c.it.kbeg;
      return true;
    }
    return false;
  };
  // piece i of this wavefront: q = wave + NW i of the stage = block row q >> 2 (A first, then the 4 of B), (k block, plane) =
  // q & 3 = wave & 3.  Eight wavefronts: pieces 0..3 are A block rows 2 i + (wave >> 2), pieces 4, 5 B block rows 2 (i - 4) +
  // (wave >> 2); four wavefronts: pieces 0..3 A block rows i, pieces 4..7 B block rows i - 4.
  constexpr int RSTEP = G::NW / 4;                          // block rows between consecutive pieces of a wavefront
  const int rb_in_tile = wave >> 2, rest_bytes = (wave & 3) * 1024;
  const char *const a_base = p.a_planes + rest_bytes + lane * 16, *const b_base = p.b_planes + rest_bytes + lane * 16;
  const int64_t row_bytes = (int64_t)p.kb16 * 2048;     // bytes of one block row of an operand: KB16 blocks x 2 planes x 1 KiB
  auto issue_piece = [&](const Cursor &c, int buf, int i) __attribute__((always_inline)) {
    const int q = wave + G::NW * i;
    const bool is_b = i >= 4;                                // (compile-time per call site)
    const int rb = is_b ? min((c.it.bn0 >> 5) + RSTEP * (i - 4) + rb_in_tile, p.b_rb_last)
                        : min((c.it.bm0 >> 5) + RSTEP * i + rb_in_tile, p.a_rb_last);
    const char *g = (is_b ? b_base : a_base) + (int64_t)rb * row_bytes + (int64_t)(c.k0 >> 4) * 2048;
    dma16(g, smem + buf * G::STAGE_BYTES + q * 1024);
  };

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
  const uint32_t thr = dropout_threshold(p.g.dropout_p);
  const float keep_scale = 1.f / (1.f - p.g.dropout_p);
  const bool partial = p.g.slab != 0;
  // LDS byte addresses of this lane's fragment chunk in the wavefront's first A / B block row of stage buffer 0
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
  const uint32_t frag = (uint32_t)chunk_index(lane & 31, lane >> 5) * 16u;
  const uint32_t va0 = lds0 + (uint32_t)(2 * wm) * G::RB_BYTES + frag, vb0 = lds0 + G::A_STAGE + (uint32_t)(2 * wn) * G::RB_BYTES + frag;

  Cursor ld = {work.begin, 0, item_at(work.begin)};
  ld.k0 = ld.it.kbeg;
  Cursor cc = ld;
#pragma unroll
  for (int i = 0; i < G::PER_WAVE; ++i) issue_piece(ld, 0, i);
  bool more_loads = advance(ld);
  int ahead = 1;                 // stages issued and not yet waited for (this one included)
  if (G::NSTAGE == 3 && more_loads) {
#pragma unroll
    for (int i = 0; i < G::PER_WAVE; ++i) issue_piece(ld, 1, i);
    more_loads = advance(ld);
    ahead = 2;
  }
  int landed = 0;                // stages whose pieces this wavefront has already waited for (in front of an epilogue)
  int buf = 0;
  for (;;) {
    // this wavefront's pieces of the stage have landed; those of the next stage may stay in flight.  (Loads complete in
    // order among themselves, so "at most 6 outstanding" implies the older six pieces are in - the tile stores of an
    // epilogue that may still be in the queue only make the wait conservative, never wrong.)
    if (landed > 0) --landed;
    else if (G::NSTAGE == 3 && ahead >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // ... and everybody's; everybody is done with the buffer that is refilled below
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t va = va0 + (uint32_t)buf * G::STAGE_BYTES, vb = vb0 + (uint32_t)buf * G::STAGE_BYTES;
    // the buffer the stage issued during this one goes to: (buf + 2) % 3, read last in stage c - 1 - or the other of two
    const int nbuf = G::NSTAGE == 3 ? (buf >= 1 ? buf - 1 : 2) : buf ^ 1;
    f16x8 fa[2][2][2], fb[2][2][2];            // [k block][tile][plane]
    PT_DS_READ_B128(fa[0][0][0], va, 0);
    PT_DS_READ_B128(fa[0][0][1], va, 1024);
    PT_DS_READ_B128(fb[0][0][0], vb, 0);
    PT_DS_READ_B128(fb[0][0][1], vb, 1024);
    PT_DS_READ_B128(fa[0][1][0], va, 4096);
    PT_DS_READ_B128(fa[0][1][1], va, 4096 + 1024);
    PT_DS_READ_B128(fb[0][1][0], vb, 4096);
    PT_DS_READ_B128(fb[0][1][1], vb, 4096 + 1024);
    PT_DS_READ_B128(fa[1][0][0], va, 2048);
    PT_DS_READ_B128(fa[1][0][1], va, 2048 + 1024);
    PT_DS_READ_B128(fb[1][0][0], vb, 2048);
    PT_DS_READ_B128(fb[1][0][1], vb, 2048 + 1024);
    PT_DS_READ_B128(fa[1][1][0], va, 4096 + 2048);
    PT_DS_READ_B128(fa[1][1][1], va, 4096 + 2048 + 1024);
    PT_DS_READ_B128(fb[1][1][0], vb, 4096 + 2048);
    PT_DS_READ_B128(fb[1][1][1], vb, 4096 + 2048 + 1024);
    const bool issue = more_loads;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (kb == 0) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pr = 0; pr < 3; ++pr) {      // smallest products first: lo hi', hi lo', hi hi'
        const int ta = pr == 0 ? 1 : 0, tb = pr == 1 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kb][i][ta], fb[kb][j][tb], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (issue) {
          issue_piece(ld, nbuf, kb * 3 + pr);
          if (G::PER_WAVE == 8 && pr == 2) issue_piece(ld, nbuf, 6 + kb);   // eight pieces in six slots
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (issue) {
      more_loads = advance(ld);
      ++ahead;
    }
    --ahead;
    if (cc.k0 + G::BK >= cc.it.kend) {  // that was the item's last stage (uniform)
      // The pieces of the next two stages (the first stages of the next tile) are waited for HERE, in front of the tile
      // stores: behind them a counted wait would also sit out the stores (2+ us when every CU writes its tile at once).
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      landed = ahead;
      float *C = p.g.C + (partial ? (size_t)cc.it.z * p.g.slab : 0);
      const int ldc = partial ? p.g.N : p.g.ldc;
      const int row0 = cc.it.bm0 + wm * 64, col0 = cc.it.bn0 + wn * 64;
      {
        const int l31 = lane & 31, lh = lane >> 5;
        float ib[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) ib[j] = inverse_of_scale(p.b_scale[min(col0 + j * 32 + l31, p.g.N - 1)]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float ia = inverse_of_scale(p.a_scale[min(row0 + i * 32 + 8 * g + 4 * lh + e, p.g.M - 1)]);
#pragma unroll
              for (int j = 0; j < 2; ++j) acc[i][j][g * 4 + e] = acc[i][j][g * 4 + e] * ia * ib[j];
            }
      }
      hp3_epilogue<EPI, KVP>(p.g, acc, C, ldc, partial, row0, col0, lane, thr, keep_scale, scratch + wave * 512, &p);
      zero_acc();
    }
    if (!advance(cc)) break;
    buf = buf == G::NSTAGE - 1 ? 0 : buf + 1;
  }
}

template <int EPI, int WM, bool KVP = false>
int launch_hp3_g(const HpParams &p, int splits, hipStream_t st) {
  using G = Hp3G<WM>;
  const int work = ((p.g.M + G::TILE_M - 1) / G::TILE_M) * ((p.g.N + HBN - 1) / HBN) * splits;
  auto kern = gemm_hp3_kernel<EPI, WM, KVP>;
  PT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS));
  const int slots = ptgemm::persistent_grid(p.g.reserved_cus) * G::WG_PER_CU;
  hipLaunchKernelGGL(kern, dim3(work < slots ? work : slots), dim3(G::THREADS), G::LDS, st, p);
  return pt_check_launch();
}
// The 256-row geometry is the one in use.  The 128-row one (two workgroups per CU, so that one's epilogue runs beside the
// other's main loop) measured the SAME per product (QKV 99.8 against 102.5 us, FFN-1 139 against 137, gated dX 146 against
// 147) and +0.1 ms in the step (profiles/r04/NOTES.md section 4): at the package power cap a better overlap buys nothing, only
// less energy per tile does.  PTAMD_HP_TILE = 128 in the environment (read at every call) selects it for measurements.
template <int EPI>
int launch_hp3(const HpParams &p, int splits, hipStream_t st) {
  const char *e = getenv("PTAMD_HP_TILE");
  return (e && e[0] == '1') ? launch_hp3_g<EPI, 2>(p, splits, st) : launch_hp3_g<EPI, 4>(p, splits, st);
}

// ---------------------------------------------------------------------------------------------- writers of the format
// scale[r] from the row maximum of a K-contiguous fp32 matrix: one wavefront per row
__global__ __launch_bounds__(256) void hp_rowscale_kernel(const float *__restrict__ x, int ld, int rows, int K, int rows_p,
                                                          float *__restrict__ scale) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows_p) return;
  if (r >= rows) {
    if (lane == 0) scale[r] = 1.f;
    return;
  }
  const float *p = x + (size_t)r * ld;
  float m = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    const float4 v = *reinterpret_cast<const float4 *>(p + k);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  m = wave_max(m);
  if (lane == 0) scale[r] = __uint_as_float(scale_bits_of(__float_as_uint(m)));
}
// transposed source ([K][rows], the operand's rows are the source's columns): column maxima by atomicMax on the bits
__global__ __launch_bounds__(256) void hp_colmax_kernel(const float *__restrict__ x, int ld, int rows, int K,
                                                        uint32_t *__restrict__ amax) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const int k0 = blockIdx.y * 64, k1 = min(K, k0 + 64);
  float m = 0.f;
  for (int k = k0; k < k1; ++k) m = fmaxf(m, fabsf(x[(size_t)k * ld + r]));
  atomicMax(amax + r, __float_as_uint(m));
}
__global__ __launch_bounds__(256) void hp_amax_to_scale_kernel(uint32_t *__restrict__ amax_scale, int rows, int rows_p) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < rows_p) amax_scale[r] = r < rows ? scale_bits_of(amax_scale[r]) : 0x3f800000u;
}
// one thread per 16-byte chunk of the padded matrix: 8 consecutive k of one row -> hi and lo
template <bool TRANSPOSED, bool SCALE_IS_AMAX>
__global__ __launch_bounds__(256) void hp_write_kernel(const float *__restrict__ x, int ld, int rows, int K, int kb16v,
                                                       const float *__restrict__ scale, char *__restrict__ planes,
                                                       int64_t nchunks) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= nchunks) return;
  const int c = (int)(id & 63);
  const int64_t blk = id >> 6;
  const int kb = (int)(blk % kb16v), rb = (int)(blk / kb16v);
  int r, h;
  chunk_coords(c, r, h);
  const int row = rb * 32 + r, k0 = kb * 16 + h * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  float s = 1.f;
  if (row < rows) {
    s = SCALE_IS_AMAX ? __uint_as_float(scale_bits_of(__float_as_uint(scale[row]))) : scale[row];
    if (!TRANSPOSED) {
      if (k0 + 8 <= K) {
        const float4 a = *reinterpret_cast<const float4 *>(x + (size_t)row * ld + k0);
        const float4 b = *reinterpret_cast<const float4 *>(x + (size_t)row * ld + k0 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (k0 + e < K) v[e] = x[(size_t)row * ld + k0 + e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (k0 + e < K) v[e] = x[(size_t)(k0 + e) * ld + row];
    }
  }
  uint4 hi, lo;
  ptsplit::split_pair_f16(v[0], v[1], s, s, hi.x, lo.x);
  ptsplit::split_pair_f16(v[2], v[3], s, s, hi.y, lo.y);
  ptsplit::split_pair_f16(v[4], v[5], s, s, hi.z, lo.z);
  ptsplit::split_pair_f16(v[6], v[7], s, s, hi.w, lo.w);
  char *dst = planes + block_offset(rb, kb, 0, kb16v) + c * 16;
  *reinterpret_cast<uint4 *>(dst) = hi;
  *reinterpret_cast<uint4 *>(dst + BLK_BYTES) = lo;
}

// ---- several K-contiguous matrices in one launch: one wavefront per (padded) row, the row held in registers
constexpr int MAX_SPLIT_JOBS = 16;
struct SplitJobs {
  const float *x[MAX_SPLIT_JOBS];
  char *planes[MAX_SPLIT_JOBS];
  float *scale[MAX_SPLIT_JOBS];
  int ld[MAX_SPLIT_JOBS], rows[MAX_SPLIT_JOBS], K[MAX_SPLIT_JOBS];
  int first_block[MAX_SPLIT_JOBS + 1];   // blocks of 4 rows
  int njobs;
};
template <int NV>
__global__ __launch_bounds__(256) void hp_split_rows_kernel(const SplitJobs j) {
  int job = 0;
#pragma unroll 1
  while (job + 1 < j.njobs && (int)blockIdx.x >= j.first_block[job + 1]) ++job;
  const int lane = threadIdx.x & 63;
  const int row = ((int)blockIdx.x - j.first_block[job]) * 4 + (threadIdx.x >> 6);
  const int rows = j.rows[job], K = j.K[job], rows_p = round_up(rows, 32), Kp = round_up(K, 32);
  if (row >= rows_p) return;
  float4 v[NV];
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    v[i] = (row < rows && c < K) ? *reinterpret_cast<const float4 *>(j.x[job] + (size_t)row * j.ld[job] + c)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[i].x), fabsf(v[i].y))), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
  }
  m = wave_max(m);
  const float s = row < rows ? __uint_as_float(scale_bits_of(__float_as_uint(m))) : 1.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < Kp) store4_split(j.planes[job], Kp >> 4, row, c, v[i], s);
  }
  if (lane == 0) j.scale[job][row] = s;
}

// ---- several ROW-contiguous matrices [K][rows] (operand rows = source columns: the weights of the dX products, B = W^T)
// with the operand rows' scales GIVEN (the column scales ptamd_weight_scales computes every step anyway): one thread per
// 16-byte chunk of the planes, one launch for all of them
struct SplitColsJobs {
  const float *x[MAX_SPLIT_JOBS];
  char *planes[MAX_SPLIT_JOBS];
  const float *scale[MAX_SPLIT_JOBS];
  int ld[MAX_SPLIT_JOBS], rows[MAX_SPLIT_JOBS], K[MAX_SPLIT_JOBS];
  int first_block[MAX_SPLIT_JOBS + 1];   // blocks of 256 chunks
  int njobs;
};
__global__ __launch_bounds__(256) void hp_split_cols_kernel(const SplitColsJobs j) {
  int job = 0;
#pragma unroll 1
  while (job + 1 < j.njobs && (int)blockIdx.x >= j.first_block[job + 1]) ++job;
  const int rows = j.rows[job], K = j.K[job], kbv = kb16(K), ld = j.ld[job];
  const int64_t nchunks = (int64_t)(round_up(rows, 32) / 32) * kbv * 64;
  const int64_t id = (int64_t)((int)blockIdx.x - j.first_block[job]) * 256 + threadIdx.x;
  if (id >= nchunks) return;
  const int c = (int)(id & 63);
  const int64_t blk = id >> 6;
  const int kb = (int)(blk % kbv), rb = (int)(blk / kbv);
  int r, h;
  chunk_coords(c, r, h);
  const int row = rb * 32 + r, k0 = kb * 16 + h * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  float sc = 1.f;
  if (row < rows) {
    sc = j.scale[job][row];
    const float *x = j.x[job];
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (k0 + e < K) v[e] = x[(size_t)(k0 + e) * ld + row];
  }
  uint4 hi, lo;
  ptsplit::split_pair_f16(v[0], v[1], sc, sc, hi.x, lo.x);
  ptsplit::split_pair_f16(v[2], v[3], sc, sc, hi.y, lo.y);
  ptsplit::split_pair_f16(v[4], v[5], sc, sc, hi.z, lo.z);
  ptsplit::split_pair_f16(v[6], v[7], sc, sc, hi.w, lo.w);
  char *dst = j.planes[job] + block_offset(rb, kb, 0, kbv) + c * 16;
  *reinterpret_cast<uint4 *>(dst) = hi;
  *reinterpret_cast<uint4 *>(dst + BLK_BYTES) = lo;
}

size_t slab_bytes(int M, int N, int split_k) {
  if (split_k <= 1) return 0;
  return ((size_t)split_k * M * N * sizeof(float) + 15) & ~(size_t)15;
}

}  // namespace
}  // namespace pthp

using namespace pthp;

extern "C" {

size_t ptamd_hp_bytes(int rows, int K) { return (rows <= 0 || K <= 0) ? 0 : plane_bytes(rows, K); }
int ptamd_hp_padded_rows(int rows) { return rows <= 0 ? 0 : round_up(rows, 32); }

int ptamd_hp_split(const float *x, int ld, int rows, int K, int transposed, void *planes, float *scale, void *stream) {
  if (rows <= 0 || K <= 0 || !x || !planes || !scale) return PTAMD_ERR_BAD_SHAPE;
  if (!transposed && ((ld & 3) || (K & 3) || !pt_aligned16(x))) return PTAMD_ERR_ALIGN;
  if (!pt_aligned16(planes)) return PTAMD_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int rows_p = round_up(rows, 32), kbv = kb16(K);
  const int64_t nchunks = (int64_t)(rows_p / 32) * kbv * 64;
  const unsigned wblocks = (unsigned)((nchunks + 255) / 256);
  if (!transposed) {
    hipLaunchKernelGGL(hp_rowscale_kernel, dim3((rows_p + 3) / 4), dim3(256), 0, st, x, ld, rows, K, rows_p, scale);
    hipLaunchKernelGGL((hp_write_kernel<false, false>), dim3(wblocks), dim3(256), 0, st, x, ld, rows, K, kbv, scale,
                       static_cast<char *>(planes), nchunks);
  } else {
    PT_HIP_TRY(hipMemsetAsync(scale, 0, (size_t)rows_p * sizeof(float), st));
    hipLaunchKernelGGL(hp_colmax_kernel, dim3((rows + 255) / 256, (K + 63) / 64), dim3(256), 0, st, x, ld, rows, K,
                       reinterpret_cast<uint32_t *>(scale));
    hipLaunchKernelGGL((hp_write_kernel<true, true>), dim3(wblocks), dim3(256), 0, st, x, ld, rows, K, kbv, scale,
                       static_cast<char *>(planes), nchunks);
    hipLaunchKernelGGL(hp_amax_to_scale_kernel, dim3((rows_p + 255) / 256), dim3(256), 0, st, reinterpret_cast<uint32_t *>(scale),
                       rows, rows_p);
  }
  return pt_check_launch();
}

int ptamd_hp_split_rows(const ptamd_hp_split_job *jobs, int njobs, void *stream) {
  if (!jobs || njobs <= 0 || njobs > MAX_SPLIT_JOBS) return PTAMD_ERR_BAD_SHAPE;
  SplitJobs j;
  int blocks = 0, kmax = 0;
  for (int i = 0; i < njobs; ++i) {
    const ptamd_hp_split_job &q = jobs[i];
    if (!q.x || !q.planes || !q.scale || q.rows <= 0 || q.K <= 0 || (q.K & 3) || (q.ld & 3) || q.K > 2048) return PTAMD_ERR_BAD_SHAPE;
    if (!pt_aligned16(q.x) || !pt_aligned16(q.planes)) return PTAMD_ERR_ALIGN;
    j.x[i] = q.x; j.planes[i] = static_cast<char *>(q.planes); j.scale[i] = q.scale;
    j.ld[i] = q.ld; j.rows[i] = q.rows; j.K[i] = q.K;
    j.first_block[i] = blocks;
    blocks += round_up(q.rows, 32) / 4;
    kmax = q.K > kmax ? q.K : kmax;
  }
  for (int i = njobs; i <= MAX_SPLIT_JOBS; ++i) j.first_block[i] = blocks;
  j.njobs = njobs;
  hipStream_t st = (hipStream_t)stream;
  if (kmax <= 256) hipLaunchKernelGGL(hp_split_rows_kernel<1>, dim3(blocks), dim3(256), 0, st, j);
  else if (kmax <= 512) hipLaunchKernelGGL(hp_split_rows_kernel<2>, dim3(blocks), dim3(256), 0, st, j);
  else if (kmax <= 1024) hipLaunchKernelGGL(hp_split_rows_kernel<4>, dim3(blocks), dim3(256), 0, st, j);
  else hipLaunchKernelGGL(hp_split_rows_kernel<8>, dim3(blocks), dim3(256), 0, st, j);
  return pt_check_launch();
}

int ptamd_hp_split_cols(const ptamd_hp_split_job *jobs, int njobs, void *stream) {
  if (!jobs || njobs <= 0 || njobs > MAX_SPLIT_JOBS) return PTAMD_ERR_BAD_SHAPE;
  SplitColsJobs j;
  int blocks = 0;
  for (int i = 0; i < njobs; ++i) {
    const ptamd_hp_split_job &q = jobs[i];
    if (!q.x || !q.planes || !q.scale || q.rows <= 0 || q.K <= 0 || q.ld < q.rows) return PTAMD_ERR_BAD_SHAPE;
    if (!pt_aligned16(q.planes)) return PTAMD_ERR_ALIGN;
    j.x[i] = q.x; j.planes[i] = static_cast<char *>(q.planes); j.scale[i] = q.scale;
    j.ld[i] = q.ld; j.rows[i] = q.rows; j.K[i] = q.K;
    j.first_block[i] = blocks;
    const int64_t nchunks = (int64_t)(round_up(q.rows, 32) / 32) * kb16(q.K) * 64;
    blocks += (int)((nchunks + 255) / 256);
  }
  for (int i = njobs; i <= MAX_SPLIT_JOBS; ++i) j.first_block[i] = blocks;
  j.njobs = njobs;
  hipLaunchKernelGGL(hp_split_cols_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, j);
  return pt_check_launch();
}

size_t ptamd_gemm_hp_workspace_bytes(int M, int N, int split_k) { return (M <= 0 || N <= 0) ? 0 : slab_bytes(M, N, split_k); }

int ptamd_gemm_hp(const ptamd_gemm_hp_args *a, void *stream) {
  if (!a || a->M <= 0 || a->N <= 0 || a->K <= 0) return PTAMD_ERR_BAD_SHAPE;
  if (!a->A || !a->B || !a->A_scale || !a->B_scale || !a->C) return PTAMD_ERR_BAD_SHAPE;
  if (!pt_aligned16(a->A) || !pt_aligned16(a->B)) return PTAMD_ERR_ALIGN;
  if (a->dropout_p < 0.f || a->dropout_p >= 1.f) return PTAMD_ERR_BAD_SHAPE;
  if ((a->flags & PTAMD_EPI_GATE) && (a->residual == nullptr) == (a->gate_mask == nullptr)) return PTAMD_ERR_BAD_SHAPE;
  if (a->gate_mask && !(a->flags & PTAMD_EPI_GATE)) return PTAMD_ERR_BAD_SHAPE;
  HpParams p;
  GemmParams &g = p.g;
  g.M = a->M; g.N = a->N; g.K = a->K;
  g.A = nullptr; g.lda = 0; g.B = nullptr; g.ldb = 0; g.C = a->C; g.ldc = a->ldc;
  g.bias = a->bias; g.residual = a->residual; g.ldr = a->ldr; g.flags = a->flags;
  g.dropout_p = a->dropout_p; g.seed = a->seed; g.stream_id = a->stream_id; g.gate_scale = a->gate_scale;
  g.reserved_cus = a->reserved_cus;
  g.colsum = nullptr; g.colsum_share = 1; g.scale_a = g.scale_b = nullptr; g.scale_a_stride = g.scale_b_stride = 1;
  g.gate_mask = a->gate_mask; g.gate_mask_out = a->gate_mask_out;
  g.mask_rb = (a->M + 31) / 32; g.mask_cb = (a->N + 31) / 32;
  constexpr int HBK = 32;
  const int Kp = round_up(a->K, 32), stages = Kp / HBK;
  int splits = a->split_k > 1 ? a->split_k : 1;
  if (splits > stages) splits = stages;
  g.k_per_split = ((stages + splits - 1) / splits) * HBK;
  splits = (Kp + g.k_per_split - 1) / g.k_per_split;
  g.splits = splits;
  g.vec_epilogue = !(a->N & 3) && !(a->ldc & 3) && pt_aligned16(a->C) &&
                   (!a->residual || (!(a->ldr & 3) && pt_aligned16(a->residual))) && (splits == 1 || pt_aligned16(a->workspace));
  g.slab = 0;
  float *user_c = a->C;
  if (splits > 1) {
    if (!a->workspace || a->workspace_bytes < slab_bytes(a->M, a->N, splits)) return PTAMD_ERR_WORKSPACE;
    g.slab = (size_t)a->M * a->N;
    g.C = static_cast<float *>(a->workspace);
  }
  p.a_planes = static_cast<const char *>(a->A);
  p.b_planes = static_cast<const char *>(a->B);
  p.a_scale = a->A_scale;
  p.b_scale = a->B_scale;
  p.kb16 = Kp / 16;
  p.a_rb_last = (round_up(a->M, 32) / 32) - 1;
  p.b_rb_last = (round_up(a->N, 32) / 32) - 1;
  p.kv_planes = static_cast<char *>(a->kv_planes);
  p.kv_inv = a->kv_inv;
  p.kv_col0 = a->kv_col0;
  p.kv_heads = a->kv_heads;
  p.kv_nt = a->M / 32;
  hipStream_t st = (hipStream_t)stream;
  if (a->kv_planes) {
    // K / V leave as planes (kv_format.h): whole 32-token tiles, 64-column heads, the plain bias epilogue of an unsplit product
    if (!a->kv_inv || a->kv_heads <= 0 || (a->M & 31) || (a->kv_col0 & 63) || a->kv_col0 < 0 ||
        a->N != a->kv_col0 + 2 * a->kv_heads * 64 || !g.vec_epilogue || splits > 1 || a->residual || a->gate_mask ||
        a->gate_mask_out || a->dropout_p != 0.f || a->flags != 0)
      return PTAMD_ERR_BAD_SHAPE;
    if (!pt_aligned16(a->kv_planes)) return PTAMD_ERR_ALIGN;
    return launch_hp3_g<ptgemm::EPI_FULL, 4, true>(p, 1, st);
  }
  const bool plain = !a->bias && !a->residual && !a->gate_mask && !a->gate_mask_out &&
                     !(a->flags & (PTAMD_EPI_RELU | PTAMD_EPI_TANH | PTAMD_EPI_ACCUM)) && a->dropout_p == 0.f;
  // the 1-bit gate (read or written) lives in the float4 epilogue of the three-stage kernel and in unsplit products only
  if ((a->gate_mask || a->gate_mask_out) && (!g.vec_epilogue || splits > 1)) return PTAMD_ERR_BAD_SHAPE;
  int rc;
  if (plain || splits > 1) rc = launch_hp<ptgemm::EPI_PLAIN>(p, splits, st);
  else if (a->dropout_p == 0.f) rc = launch_hp<ptgemm::EPI_NODROP>(p, splits, st);
  else rc = launch_hp<ptgemm::EPI_FULL>(p, splits, st);
  if (rc || splits == 1) return rc;
  const float *slabs = g.C;
  g.C = user_c;
  return ptgemm::launch_splitk_reduce(g, slabs, splits, nullptr, nullptr, st);
}

}  // extern "C"
