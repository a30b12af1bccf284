// ptamd_gemm in f16x2 arithmetic: the 3-product instantiations of gemm_split_kernel.h and the pass over the operands
// that finds the row scales (include/ptamd.h, PTAMD_GEMM_F16X2).
#include "gemm_split_kernel.h"

namespace ptgemm {
namespace {

// ---- row scales of the f16x2 arithmetic: scale[r] = the power of two that takes max_k |x[r][k]| into [2^14, 2^15).
// One launch covers both operands (blocks [0, ja.blocks) work on A, the rest on B).
//   K-contiguous operand [rows][K]: a wavefront reduces two rows (RS_ROWS = 8 per block) and stores their scales;
//   row-contiguous operand [K][rows]: a block takes 256 rows x a chunk of RS_KCHUNK k, its threads 4 consecutive rows
//                                   each, and the chunks meet in an atomicMin on the scale bits (a larger maximum is a
//                                   smaller scale; the array is preset to the largest scale by the launcher);
//   a SMALL row-contiguous operand (a weight matrix, <= 2^21 elements): a block takes 16 rows and all of K, 64 k at a
//                                   time (4 threads x 4 rows per k), and stores the scales - no preset, no atomics.
struct ScaleJob {
  const float *x;
  int ld, rows, K, kmajor;
  uint32_t *scale;
  int blocks;
};
constexpr int RS_THREADS = 256, RS_ROWS = 8, RS_KCHUNK = 128;
constexpr int RS_KMAJOR_CHUNKS = 1, RS_KMAJOR_WHOLE = 2;  // ScaleJob::kmajor

__device__ __forceinline__ float absmax4(float m, const float4 v) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}

__global__ __launch_bounds__(RS_THREADS) void gemm_row_scale_kernel(const ScaleJob ja, const ScaleJob jb) {
  const bool second = (int)blockIdx.x >= ja.blocks;
  const ScaleJob j = second ? jb : ja;
  const int b = (int)blockIdx.x - (second ? ja.blocks : 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (!j.kmajor) {
    // RS_ROWS / 4 rows per wavefront, two at a time: many small wavefronts (this pass is latency-bound otherwise)
    for (int rr = 0; rr < RS_ROWS / 4; rr += 2) {
      const int r0 = b * RS_ROWS + wave * (RS_ROWS / 4) + rr;
      if (r0 >= j.rows) break;  // wavefront-uniform
      const int r1 = min(r0 + 1, j.rows - 1);  // (past the end: the last row again, same value, same store)
      const float *p0 = j.x + (size_t)r0 * j.ld, *p1 = j.x + (size_t)r1 * j.ld;
      float m0 = 0.f, m1 = 0.f;
#pragma unroll 2
      for (int k = lane * 4; k < j.K; k += 256) {
        m0 = absmax4(m0, *reinterpret_cast<const float4 *>(p0 + k));
        m1 = absmax4(m1, *reinterpret_cast<const float4 *>(p1 + k));
      }
#pragma unroll
      for (int o = 32; o; o >>= 1) {
        m0 = fmaxf(m0, __shfl_xor(m0, o));
        m1 = fmaxf(m1, __shfl_xor(m1, o));
      }
      if (lane == 0) {
        j.scale[r0] = row_scale_bits(__float_as_uint(m0));
        j.scale[r1] = row_scale_bits(__float_as_uint(m1));
      }
    }
  } else if (j.kmajor == RS_KMAJOR_WHOLE) {
    __shared__ float4 red[RS_THREADS];
    const int r4 = min(b * 16 + (tid & 3) * 4, j.rows - 4);  // (rows % 4 == 0; a clamped quad repeats its neighbour's work)
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int k = tid >> 2; k < j.K; k += RS_THREADS / 4) {
      const float4 v = *reinterpret_cast<const float4 *>(j.x + (size_t)k * j.ld + r4);
      m.x = fmaxf(m.x, fabsf(v.x)); m.y = fmaxf(m.y, fabsf(v.y)); m.z = fmaxf(m.z, fabsf(v.z)); m.w = fmaxf(m.w, fabsf(v.w));
    }
    red[tid] = m;
    __syncthreads();
    for (int half = RS_THREADS / 2; half >= 4; half >>= 1) {  // tid & 3 (the row quad) is preserved by every step
      if (tid < half) {
        const float4 o = red[tid + half];
        m.x = fmaxf(m.x, o.x); m.y = fmaxf(m.y, o.y); m.z = fmaxf(m.z, o.z); m.w = fmaxf(m.w, o.w);
        red[tid] = m;
      }
      __syncthreads();
    }
    if (tid < 4) {
      j.scale[r4 + 0] = row_scale_bits(__float_as_uint(m.x));
      j.scale[r4 + 1] = row_scale_bits(__float_as_uint(m.y));
      j.scale[r4 + 2] = row_scale_bits(__float_as_uint(m.z));
      j.scale[r4 + 3] = row_scale_bits(__float_as_uint(m.w));
    }
  } else {
    __shared__ float4 red[RS_THREADS];
    const int groups = (j.rows + 255) / 256;
    const int grp = b % groups, chunk = b / groups;
    const int r4 = grp * 256 + lane * 4;
    const int kbeg = chunk * RS_KCHUNK, kend = min(j.K, kbeg + RS_KCHUNK);
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r4 < j.rows) {
#pragma unroll 8
      for (int k = kbeg + wave; k < kend; k += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(j.x + (size_t)k * j.ld + r4);
        m.x = fmaxf(m.x, fabsf(v.x)); m.y = fmaxf(m.y, fabsf(v.y)); m.z = fmaxf(m.z, fabsf(v.z)); m.w = fmaxf(m.w, fabsf(v.w));
      }
    }
    red[tid] = m;
    __syncthreads();
    if (wave == 0 && r4 < j.rows) {
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float4 o = red[tid + 64 * w];
        m.x = fmaxf(m.x, o.x); m.y = fmaxf(m.y, o.y); m.z = fmaxf(m.z, o.z); m.w = fmaxf(m.w, o.w);
      }
      atomicMin(j.scale + r4 + 0, row_scale_bits(__float_as_uint(m.x)));
      atomicMin(j.scale + r4 + 1, row_scale_bits(__float_as_uint(m.y)));
      atomicMin(j.scale + r4 + 2, row_scale_bits(__float_as_uint(m.z)));
      atomicMin(j.scale + r4 + 3, row_scale_bits(__float_as_uint(m.w)));
    }
  }
}

}  // namespace

int launch_row_scales(const GemmParams &p, bool a_kmajor, bool b_kmajor, uint32_t *scale_a, uint32_t *scale_b, hipStream_t st) {
  auto job = [](const float *x, int ld, int rows, int K, bool kmajor, uint32_t *scale) {
    const int mode = !kmajor ? 0 : ((size_t)rows * K <= ((size_t)1 << 21) ? RS_KMAJOR_WHOLE : RS_KMAJOR_CHUNKS);
    ScaleJob j = {x, ld, rows, K, mode, scale, 0};
    j.blocks = mode == RS_KMAJOR_CHUNKS  ? ((rows + 255) / 256) * ((K + RS_KCHUNK - 1) / RS_KCHUNK)
               : mode == RS_KMAJOR_WHOLE ? (rows + 15) / 16
                                         : (rows + RS_ROWS - 1) / RS_ROWS;
    return j;
  };
  ScaleJob ja = job(p.A, p.lda, p.M, p.K, a_kmajor, scale_a), jb = job(p.B, p.ldb, p.N, p.K, b_kmajor, scale_b);
  if (!scale_a) ja.blocks = 0;  // provided by the caller
  if (!scale_b) jb.blocks = 0;
  if (ja.blocks + jb.blocks == 0) return PTAMD_OK;
  if (ja.blocks && ja.kmajor == RS_KMAJOR_CHUNKS)
    PT_HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(scale_a), (int)(254u << 23), (size_t)p.M, st));
  if (jb.blocks && jb.kmajor == RS_KMAJOR_CHUNKS)
    PT_HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(scale_b), (int)(254u << 23), (size_t)p.N, st));
  hipLaunchKernelGGL(gemm_row_scale_kernel, dim3(ja.blocks + jb.blocks), dim3(RS_THREADS), 0, st, ja, jb);
  return pt_check_launch();
}

int launch_group_f16x2(const GemmGroup &g, hipStream_t st) { return launch_group<3>(g, st); }

int launch_split_f16x2(const GemmParams &p, bool a_kmajor, bool b_kmajor, int splits, hipStream_t st) {
  // the smallest epilogue that does the job (instruction-cache footprint, see gemm_common.h)
  const bool plain = p.slab != 0 || (!p.bias && !p.residual && !p.flags && p.dropout_p == 0.f);
  if (plain) return launch_layout<3, EPI_PLAIN>(p, a_kmajor, b_kmajor, splits, st);
  if (p.dropout_p == 0.f) return launch_layout<3, EPI_NODROP>(p, a_kmajor, b_kmajor, splits, st);
  return launch_layout<3, EPI_FULL>(p, a_kmajor, b_kmajor, splits, st);
}

}  // namespace ptgemm
