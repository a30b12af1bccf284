// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 products and accumulation,
// 64 FLOP/clk/SIMD = 157 TF/s chip peak) with fused bias / ReLU / dropout / residual / tanh epilogues.
//
// Replaces every torch.nn.Linear of the reference encoder and their autograd backward GEMMs:
//   wq/wk/wv/wo     /root/reference/protein_transformer/models/transformer/Attention.py:38-41,49,69
//   pwff.layer1/2   .../models/transformer/Sublayers.py:28-34
//   output_projection + tanh  .../models/encoder_only.py:18,39-41
//   residual + dropout of SublayerConnection  .../models/transformer/Sublayers.py:16-17
//
// Tiling (MI355X-first, wave64): block = 128 x 128 outputs, 4 wavefronts in a 2 x 2 grid, each wavefront
// owns 64 x 64 = 2 x 2 MFMA tiles of 32 x 32 (64 accumulator VGPRs).  K advances 32 per stage through a
// double-buffered, k-major LDS image [k][row]: a lane's MFMA operand A[i = lane&31][k = lane>>5] is then a
// conflict-free ds_read_b32 at row stride 1.  The f32 MFMA needs only ONE operand dword per lane per 64
// cycles, so LDS bandwidth is irrelevant; the kernel is bound by the MFMA issue rate once the global->LDS
// staging (register double buffering, one barrier per stage) is hidden.
// Workgroup ids are remapped so that the tiles of one 128-row panel of A run on the same XCD (shared L2).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;
constexpr int LD_T = 129;  // k-major image filled by transposing K-contiguous rows: odd stride, conflict-free
constexpr int LD_C = 132;  // k-major image filled by straight 16-byte copies

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
  size_t slab;      // M*N when split-K writes partial slabs, else 0
};

// ---- staging: each thread carries 4 float4 per operand per stage
template <bool KMAJOR>
__device__ __forceinline__ void load_stage(const float *__restrict__ src, int ld, int rows, int r0, int K, int k0,
                                           int tid, float4 (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (!KMAJOR) {  // src[row][k]: 8 lanes cover 32 consecutive k of one row
      const int row = r0 + (tid >> 3) + 32 * i, k = k0 + 4 * (tid & 7);
      v[i] = (row < rows && k < K) ? *reinterpret_cast<const float4 *>(src + (size_t)row * ld + k)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {  // src[k][row]: 32 lanes cover 128 consecutive rows of one k
      const int k = k0 + (tid >> 5) + 8 * i, row = r0 + 4 * (tid & 31);
      v[i] = (k < K && row < rows) ? *reinterpret_cast<const float4 *>(src + (size_t)k * ld + row)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}
template <bool KMAJOR>
__device__ __forceinline__ void store_stage(float *__restrict__ s, int tid, const float4 (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (!KMAJOR) {
      const int row = (tid >> 3) + 32 * i, k = 4 * (tid & 7);
      s[(k + 0) * LD_T + row] = v[i].x;
      s[(k + 1) * LD_T + row] = v[i].y;
      s[(k + 2) * LD_T + row] = v[i].z;
      s[(k + 3) * LD_T + row] = v[i].w;
    } else {
      const int k = (tid >> 5) + 8 * i, row = 4 * (tid & 31);
      *reinterpret_cast<float4 *>(s + k * LD_C + row) = v[i];
    }
  }
}

__device__ __forceinline__ float epilogue_value(float v, int row, int col, const GemmParams &p, uint32_t thr,
                                                float keep_scale, const uint4 &rnd) {
  if (p.bias) v += p.bias[col];
  if (p.flags & PTAMD_EPI_RELU) v = fmaxf(v, 0.f);
  if (p.dropout_p > 0.f) {
    const uint32_t w = (row & 3) == 0 ? rnd.x : (row & 3) == 1 ? rnd.y : (row & 3) == 2 ? rnd.z : rnd.w;
    v = (w >= thr) ? v * keep_scale : 0.f;
  }
  if (p.residual) v += p.residual[(size_t)row * p.ldr + col];
  if (p.flags & PTAMD_EPI_TANH) v = tanhf(v);
  return v;
}

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(NT, 2) void gemm_f32_mfma_kernel(const GemmParams p) {
  constexpr int LDA = A_KMAJOR ? LD_C : LD_T, LDB = B_KMAJOR ? LD_C : LD_T;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *const sA0 = smem;                 // two stages of A, then two stages of B
  float *const sB0 = smem + 2 * BK * LDA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;

  // XCD-aware, bijective remap of the linear workgroup id: consecutive logical tiles (which share the same
  // A row panel) land on the same XCD because the dispatcher round-robins workgroup b to XCD b % 8.
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int nwg = tiles_m * tiles_n;
  int logical;
  {
    const int id = blockIdx.x, xcd = id & 7, q = nwg >> 3, r = nwg & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int bm0 = (logical / tiles_n) * BM, bn0 = (logical % tiles_n) * BN;
  const int kbeg = blockIdx.z * p.k_per_split, kend = min(p.K, kbeg + p.k_per_split);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[4], rb[4];
  load_stage<A_KMAJOR>(p.A, p.lda, p.M, bm0, kend, kbeg, tid, ra);
  load_stage<B_KMAJOR>(p.B, p.ldb, p.N, bn0, kend, kbeg, tid, rb);
  store_stage<A_KMAJOR>(sA0, tid, ra);
  store_stage<B_KMAJOR>(sB0, tid, rb);
  __syncthreads();

  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) {  // next stage's global loads fly under this stage's MFMAs
      load_stage<A_KMAJOR>(p.A, p.lda, p.M, bm0, kend, k0 + BK, tid, ra);
      load_stage<B_KMAJOR>(p.B, p.ldb, p.N, bn0, kend, k0 + BK, tid, rb);
    }
    const float *a_base = sA0 + cur * (BK * LDA) + lh * LDA + wm * 64 + l31;
    const float *b_base = sB0 + cur * (BK * LDB) + lh * LDB + wn * 64 + l31;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const float a0 = a_base[(2 * kk) * LDA], a1 = a_base[(2 * kk) * LDA + 32];
      const float b0 = b_base[(2 * kk) * LDB], b1 = b_base[(2 * kk) * LDB + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) {
      store_stage<A_KMAJOR>(sA0 + (cur ^ 1) * (BK * LDA), tid, ra);
      store_stage<B_KMAJOR>(sB0 + (cur ^ 1) * (BK * LDB), tid, rb);
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const bool partial = p.slab != 0;
  float *C = p.C + (partial ? (size_t)blockIdx.z * p.slab : 0);
  const uint32_t thr = dropout_threshold(p.dropout_p);
  const float keep_scale = 1.f / (1.f - p.dropout_p);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = bn0 + wn * 64 + j * 32 + l31;
      if (col >= p.N) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int rowq = bm0 + wm * 64 + i * 32 + 8 * g + 4 * lh;  // 4 consecutive rows share one Philox call
        uint4 rnd = make_uint4(0, 0, 0, 0);
        if (!partial && p.dropout_p > 0.f) rnd = philox4x32(p.seed, (uint64_t)(rowq >> 2) * p.N + col, p.stream_id);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = rowq + e;
          if (row >= p.M) continue;
          float v = acc[i][j][g * 4 + e];
          if (!partial) {
            v = epilogue_value(v, row, col, p, thr, keep_scale, rnd);
            if (p.flags & PTAMD_EPI_ACCUM) v += C[(size_t)row * p.ldc + col];
            C[(size_t)row * p.ldc + col] = v;
          } else {
            C[(size_t)row * p.N + col] = v;
          }
        }
      }
    }
}

// split-K: sum the slabs in a fixed order, then the same epilogue
__global__ void gemm_splitk_reduce_kernel(const GemmParams p, const float *__restrict__ slabs, int splits) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int rowq = blockIdx.y * 4;
  if (col >= p.N) return;
  const uint32_t thr = dropout_threshold(p.dropout_p);
  const float keep_scale = 1.f / (1.f - p.dropout_p);
  uint4 rnd = make_uint4(0, 0, 0, 0);
  if (p.dropout_p > 0.f) rnd = philox4x32(p.seed, (uint64_t)(rowq >> 2) * p.N + col, p.stream_id);
  for (int e = 0; e < 4; ++e) {
    const int row = rowq + e;
    if (row >= p.M) return;
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += slabs[(size_t)s * p.slab + (size_t)row * p.N + col];
    v = epilogue_value(v, row, col, p, thr, keep_scale, rnd);
    if (p.flags & PTAMD_EPI_ACCUM) v += p.C[(size_t)row * p.ldc + col];
    p.C[(size_t)row * p.ldc + col] = v;
  }
}

template <bool AK, bool BK_>
int launch(const GemmParams &p, int splits, hipStream_t st) {
  constexpr int LDA = AK ? LD_C : LD_T, LDB = BK_ ? LD_C : LD_T;
  const size_t lds = (size_t)2 * BK * (LDA + LDB) * sizeof(float);
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  auto kern = gemm_f32_mfma_kernel<AK, BK_>;
  static bool attr_set = false;
  if (!attr_set) {
    PT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles, 1, splits), dim3(NT), lds, st, p);
  return pt_check_launch();
}

}  // namespace

extern "C" {

size_t ptamd_gemm_workspace_bytes(int M, int N, int split_k) {
  if (split_k <= 1 || M <= 0 || N <= 0) return 0;
  return (size_t)split_k * M * N * sizeof(float);
}

int ptamd_gemm(const ptamd_gemm_args *a, void *stream) {
  if (!a || a->M <= 0 || a->N <= 0 || a->K <= 0) return PTAMD_ERR_BAD_SHAPE;
  if ((a->K & 3) || (a->lda & 3) || (a->ldb & 3)) return PTAMD_ERR_BAD_SHAPE;
  if (a->a_kmajor && (a->M & 3)) return PTAMD_ERR_BAD_SHAPE;
  if (a->b_kmajor && (a->N & 3)) return PTAMD_ERR_BAD_SHAPE;
  if (!pt_aligned16(a->A) || !pt_aligned16(a->B)) return PTAMD_ERR_ALIGN;
  if (a->dropout_p < 0.f || a->dropout_p >= 1.f) return PTAMD_ERR_BAD_SHAPE;
  int splits = a->split_k > 1 ? a->split_k : 1;
  const int kblocks = (a->K + BK - 1) / BK;
  if (splits > kblocks) splits = kblocks;
  GemmParams p;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.A = a->A; p.lda = a->lda; p.B = a->B; p.ldb = a->ldb; p.C = a->C; p.ldc = a->ldc;
  p.bias = a->bias; p.residual = a->residual; p.ldr = a->ldr; p.flags = a->flags;
  p.dropout_p = a->dropout_p; p.seed = a->seed; p.stream_id = a->stream_id;
  p.k_per_split = ((kblocks + splits - 1) / splits) * BK;
  splits = (a->K + p.k_per_split - 1) / p.k_per_split;
  p.slab = 0;
  float *user_c = a->C;
  if (splits > 1) {
    if (!a->workspace || a->workspace_bytes < (size_t)splits * a->M * a->N * sizeof(float)) return PTAMD_ERR_WORKSPACE;
    p.slab = (size_t)a->M * a->N;
    p.C = static_cast<float *>(a->workspace);
  }
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (!a->a_kmajor && !a->b_kmajor) rc = launch<false, false>(p, splits, st);
  else if (!a->a_kmajor && a->b_kmajor) rc = launch<false, true>(p, splits, st);
  else if (a->a_kmajor && !a->b_kmajor) rc = launch<true, false>(p, splits, st);
  else rc = launch<true, true>(p, splits, st);
  if (rc || splits == 1) return rc;
  const float *slabs = p.C;
  p.C = user_c;
  hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((a->N + 255) / 256, (a->M + 3) / 4), dim3(256), 0, st, p, slabs,
                     splits);
  return pt_check_launch();
}

}  // extern "C"
