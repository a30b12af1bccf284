// Per-step bookkeeping of the f16x2 GEMM arithmetic (include/ptamd.h, PTAMD_GEMM_F16X2) that does NOT need a pass over
// the activations: the row / column scales of the weight matrices, and upper bounds of the row maxima of activations
// that follow from the weights alone.
//
// ptamd_gemm in f16x2 arithmetic wants, per operand row, a power of two that takes the row's largest |x| just below
// 2^15.  Left to itself it finds them with a pass over both operands in front of every product (gemm_row_scale_kernel:
// 0.72 ms of the 15 ms benchmark step).  Instead:
//   * weights change once per step: ONE launch here computes, for every weight matrix of the model, the scale of every
//     row (the B operand of the forward products x W^T) and of every column (the B operand of the backward products
//     dy W), plus the largest row / column L2 norm and the largest |w| of the matrix;
//   * activations written by row-wise kernels get their exact scale from that kernel (LayerNorm forward; LayerNorm
//     backward + dropout);
//   * activations written tile-wise (attention output, ReLU(x W1^T + b1), dy W2) get a BOUND instead of the maximum:
//       |LN(x)|_2 <= max|gamma| sqrt(D) + |beta|_2                      (the normalised row has |z|_2 <= sqrt(D))
//       |x W^T + b|_inf <= |x|_2 max_n |W[n]|_2 + max|b|                (Cauchy-Schwarz)
//       |softmax-average of V rows| <= max |V|;  ReLU and the dropout mask only shrink;  / (1 - p) for the dropout scale
//     A bound that is 2^k above the true maximum costs k of the 18 binades in which an element keeps its full 22 bits
//     (measured 2-5 on the benchmark model, tests/test_gpu_scales.py), nothing else: the scale stays a power of two, the
//     products stay exact.  ptamd_bound_scales turns the statistics of the weight launch into those scales on the device.
// No host synchronisation anywhere; both entry points take their (small) job lists by value in the kernel arguments.
#include "common.h"

namespace {

constexpr int MAX_JOBS = 40;
struct WJobs {
  ptamd_wscale_job job[MAX_JOBS];
  int first_block[MAX_JOBS + 1];  // blocks [first_block[j], first_block[j+1]) work on job j: row blocks, then column blocks
  int row_blocks[MAX_JOBS];
  int n;
};

__global__ void wscale_init_kernel(const WJobs js) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < js.n && js.job[j].stats) {
    float *s = js.job[j].stats;
    s[0] = s[1] = s[2] = s[3] = 0.f;
  }
}

// rows: 32 per block (a wavefront per row, two rows each, 16-byte loads where the row allows them);
// columns: 64 per block over all rows (16 x 64 threads, a float4 of columns per thread, 8 rows in flight per thread, LDS
// tree).  1024 threads: a column block of a [2048, 512] matrix is the launch's critical path - 128 dependent-looking rounds
// of loads per thread at 256 threads (47 us for the model's weights), 32 at 1024.
constexpr int COLS_PER_BLOCK = 64, ROWS_PER_BLOCK = 32, WS_THREADS = 1024, WS_WAVES = WS_THREADS / 64, WS_RY = WS_THREADS / 16;
__global__ __launch_bounds__(WS_THREADS) void wscale_kernel(const WJobs js) {
  int j = 0;
  while (j + 1 < js.n && (int)blockIdx.x >= js.first_block[j + 1]) ++j;
  const ptamd_wscale_job jb = js.job[j];
  const int b = (int)blockIdx.x - js.first_block[j];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t *stat_bits = reinterpret_cast<uint32_t *>(jb.stats);
  const bool vec = !(jb.cols & 3) && !(jb.ld & 3) && !(reinterpret_cast<uintptr_t>(jb.w) & 15);
  if (b < js.row_blocks[j]) {
    // 32 rows per block, 2 per wavefront; ONE pair of atomics per block (atomics on one address serialise in the L2: a pair
    // per row made this the slowest part of the kernel)
    __shared__ float s_nrm[WS_WAVES], s_amx[WS_WAVES];
    float bn = 0.f, bm = 0.f;
    for (int rr = 0; rr < ROWS_PER_BLOCK / WS_WAVES; ++rr) {
      const int r = b * ROWS_PER_BLOCK + wave * (ROWS_PER_BLOCK / WS_WAVES) + rr;
      if (r >= jb.rows) break;  // wavefront-uniform
      const float *p = jb.w + (size_t)r * jb.ld;
      float m = 0.f, sq = 0.f;
      if (vec) {
        for (int c = lane * 4; c < jb.cols; c += 256) {
          const float4 v = *reinterpret_cast<const float4 *>(p + c);
          m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
          sq = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, sq))));
        }
      } else {
        for (int c = lane; c < jb.cols; c += 64) {
          const float v = p[c];
          m = fmaxf(m, fabsf(v));
          sq = fmaf(v, v, sq);
        }
      }
      m = wave_max(m);
      sq = wave_sum(sq);  // (fixed order: deterministic)
      if (lane == 0 && jb.row_scale) jb.row_scale[r] = pt_row_scale_bits(__float_as_uint(m));
      bn = fmaxf(bn, sqrtf(sq));
      bm = fmaxf(bm, m);
    }
    if (stat_bits) {
      if (lane == 0) {
        s_nrm[wave] = bn;
        s_amx[wave] = bm;
      }
      __syncthreads();
      if (tid == 0) {  // non-negative floats order like their bit patterns
        float n = 0.f, a = 0.f;
#pragma unroll
        for (int k = 0; k < WS_WAVES; ++k) {
          n = fmaxf(n, s_nrm[k]);
          a = fmaxf(a, s_amx[k]);
        }
        atomicMax(stat_bits + 0, __float_as_uint(n));
        atomicMax(stat_bits + 2, __float_as_uint(a));
      }
    }
  } else {
    // (column sums of squares in fp64: the largest column norm is then the same fp32 number whatever the order of the sum -
    // csrc/wprep.hip sums per 32-row block - and with it every bound derived from it)
    __shared__ float4 s_m[WS_RY][17];
    __shared__ double s_q[WS_RY][17][4];
    const int cb = b - js.row_blocks[j];
    const int cx = tid & 15, ry = tid >> 4, c = cb * COLS_PER_BLOCK + cx * 4;
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
    double sq[4] = {0.0, 0.0, 0.0, 0.0};
    auto take = [&](const float4 v) __attribute__((always_inline)) {
      m.x = fmaxf(m.x, fabsf(v.x)); m.y = fmaxf(m.y, fabsf(v.y)); m.z = fmaxf(m.z, fabsf(v.z)); m.w = fmaxf(m.w, fabsf(v.w));
      sq[0] = fma((double)v.x, (double)v.x, sq[0]); sq[1] = fma((double)v.y, (double)v.y, sq[1]);
      sq[2] = fma((double)v.z, (double)v.z, sq[2]); sq[3] = fma((double)v.w, (double)v.w, sq[3]);
    };
    if (c < jb.cols) {
      if (vec) {
#pragma unroll 8
        for (int r = ry; r < jb.rows; r += WS_RY) take(*reinterpret_cast<const float4 *>(jb.w + (size_t)r * jb.ld + c));
      } else {
        for (int r = ry; r < jb.rows; r += WS_RY) {
          const float *q = jb.w + (size_t)r * jb.ld + c;
          take(make_float4(q[0], c + 1 < jb.cols ? q[1] : 0.f, c + 2 < jb.cols ? q[2] : 0.f, c + 3 < jb.cols ? q[3] : 0.f));
        }
      }
    }
    s_m[ry][cx] = m;
#pragma unroll
    for (int e = 0; e < 4; ++e) s_q[ry][cx][e] = sq[e];
    __syncthreads();
    if (ry == 0 && c < jb.cols) {   // (fixed order: deterministic)
#pragma unroll 8
      for (int k = 1; k < WS_RY; ++k) {
        const float4 a = s_m[k][cx];
        m.x = fmaxf(m.x, a.x); m.y = fmaxf(m.y, a.y); m.z = fmaxf(m.z, a.z); m.w = fmaxf(m.w, a.w);
#pragma unroll
        for (int e = 0; e < 4; ++e) sq[e] += s_q[k][cx][e];
      }
      const float mm[4] = {m.x, m.y, m.z, m.w};
      float nmax = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < jb.cols) {
          if (jb.col_scale) jb.col_scale[c + e] = pt_row_scale_bits(__float_as_uint(mm[e]));
          nmax = fmaxf(nmax, (float)sqrt(sq[e]));
        }
      if (stat_bits) atomicMax(stat_bits + 1, __float_as_uint(nmax));
    }
  }
}

struct BJobs {
  ptamd_bound_job job[MAX_JOBS];
  int n;
};
__global__ void bound_kernel(const BJobs js) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= js.n) return;
  const ptamd_bound_job b = js.job[j];
  float x = 1.f;  // bound of the L2 norm of the input row
  if (b.ln_gamma_stats) x = b.ln_gamma_stats[2] * b.sqrt_d + (b.ln_beta_stats ? b.ln_beta_stats[0] : 0.f);
  float v = x * b.w_stats[b.w_stat_index];
  if (b.bias_stats) v += b.bias_stats[2];
  v *= b.post_scale;
  if (b.out_scale) b.out_scale[0] = b.out_scale[1] = b.out_scale[2] = b.out_scale[3] = pt_row_scale_bits(__float_as_uint(v));
  if (b.out_value) *b.out_value = v;
}

// presets of small device words (atomicMin targets, flags) of up to 8 buffers in one launch
constexpr int MAX_FILLS = 8, FILL_THREADS = 256;
struct FJobs {
  ptamd_fill_job job[MAX_FILLS];
  int first_block[MAX_FILLS + 1];
  int n;
};
__global__ __launch_bounds__(FILL_THREADS) void fill_u32_kernel(const FJobs js) {
  int j = 0;
#pragma unroll
  for (int k = 1; k < MAX_FILLS; ++k) j += (k < js.n && (int)blockIdx.x >= js.first_block[k]) ? 1 : 0;
  const ptamd_fill_job f = js.job[j];
  const int64_t i = ((int64_t)(blockIdx.x - js.first_block[j]) * FILL_THREADS + threadIdx.x) * 4;
  if (i + 4 <= f.n && (reinterpret_cast<uintptr_t>(f.dst) & 15) == 0) {
    *reinterpret_cast<uint4 *>(f.dst + i) = make_uint4(f.value, f.value, f.value, f.value);
  } else {
    for (int64_t k = i; k < f.n && k < i + 4; ++k) f.dst[k] = f.value;
  }
}

}  // namespace

extern "C" {

int ptamd_fill_u32(const ptamd_fill_job *jobs, int njobs, void *stream) {
  if (!jobs || njobs <= 0 || njobs > MAX_FILLS) return PTAMD_ERR_BAD_SHAPE;
  FJobs js;
  js.n = njobs;
  int blocks = 0;
  for (int j = 0; j < njobs; ++j) {
    if (!jobs[j].dst || jobs[j].n <= 0) return PTAMD_ERR_BAD_SHAPE;
    js.job[j] = jobs[j];
    js.first_block[j] = blocks;
    blocks += (int)((jobs[j].n + 4 * FILL_THREADS - 1) / (4 * FILL_THREADS));
  }
  js.first_block[njobs] = blocks;
  hipLaunchKernelGGL(fill_u32_kernel, dim3(blocks), dim3(FILL_THREADS), 0, (hipStream_t)stream, js);
  return pt_check_launch();
}

int ptamd_weight_scales(const ptamd_wscale_job *jobs, int njobs, void *stream) {
  if (!jobs || njobs <= 0 || njobs > MAX_JOBS) return PTAMD_ERR_BAD_SHAPE;
  WJobs js;
  js.n = njobs;
  int blocks = 0;
  for (int j = 0; j < njobs; ++j) {
    const ptamd_wscale_job &q = jobs[j];
    if (!q.w || q.rows <= 0 || q.cols <= 0 || q.ld < q.cols) return PTAMD_ERR_BAD_SHAPE;
    js.job[j] = q;
    js.first_block[j] = blocks;
    js.row_blocks[j] = (q.row_scale || q.stats) ? (q.rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK : 0;
    blocks += js.row_blocks[j] + (((q.col_scale || q.stats) && !q.rows_only) ? (q.cols + COLS_PER_BLOCK - 1) / COLS_PER_BLOCK : 0);
  }
  js.first_block[njobs] = blocks;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(wscale_init_kernel, dim3(1), dim3(64), 0, st, js);
  if (blocks > 0) hipLaunchKernelGGL(wscale_kernel, dim3(blocks), dim3(WS_THREADS), 0, st, js);
  return pt_check_launch();
}

int ptamd_bound_scales(const ptamd_bound_job *jobs, int njobs, void *stream) {
  if (!jobs || njobs <= 0 || njobs > MAX_JOBS) return PTAMD_ERR_BAD_SHAPE;
  BJobs js;
  js.n = njobs;
  for (int j = 0; j < njobs; ++j) {
    if (!jobs[j].w_stats || jobs[j].w_stat_index < 0 || jobs[j].w_stat_index > 2) return PTAMD_ERR_BAD_SHAPE;
    js.job[j] = jobs[j];
  }
  hipLaunchKernelGGL(bound_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, js);
  return pt_check_launch();
}

}  // extern "C"
