// Masked MSE over the (cos, sin) angle representation, full / backbone / side-chain in one pass.
//
// Replaces mse_over_angles (/root/reference/protein_transformer/losses.py:175-214), which the reference
// calls three times per step (train.py:64-66): rows whose truth is entirely zero *within the selected
// column slice* are batch padding, NaN truth elements are missing angles, the rest enter a plain mean
// of squared differences.  Backbone = columns 0..11, side chain = columns 12..23 of the 24.
#include "common.h"

namespace {

constexpr int MB = 256;    // threads per block
constexpr int NBLK = 128;  // blocks of the first pass: one fp64 partial row of 6 sums each

// pass 1: every block strides over the rows and leaves its 6 partial sums (fp64) in part[block][6]
__global__ __launch_bounds__(MB) void mse_angles_partial_kernel(const float *__restrict__ pred,
                                                                const float *__restrict__ truth, int64_t T,
                                                                double *__restrict__ part) {
  __shared__ double s_red[MB / 64][6];
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int64_t t = (int64_t)blockIdx.x * MB + threadIdx.x; t < T; t += (int64_t)NBLK * MB) {
    const float4 *tp = reinterpret_cast<const float4 *>(truth + t * 24);
    const float4 *pp = reinterpret_cast<const float4 *>(pred + t * 24);
    float tv[24], pv[24];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      float4 a = tp[q], b = pp[q];
      tv[q * 4] = a.x; tv[q * 4 + 1] = a.y; tv[q * 4 + 2] = a.z; tv[q * 4 + 3] = a.w;
      pv[q * 4] = b.x; pv[q * 4 + 1] = b.y; pv[q * 4 + 2] = b.z; pv[q * 4 + 3] = b.w;
    }
    bool any_bb = false, any_sc = false;
    float s_bb = 0.f, s_sc = 0.f, c_bb = 0.f, c_sc = 0.f;
#pragma unroll
    for (int k = 0; k < 24; ++k) {
      const bool nz = tv[k] != 0.f;  // NaN != 0 is true, exactly like torch.ne
      const bool ok = !isnan(tv[k]);
      const float d = pv[k] - tv[k];
      if (k < 12) {
        any_bb |= nz;
        if (ok) { s_bb += d * d; c_bb += 1.f; }
      } else {
        any_sc |= nz;
        if (ok) { s_sc += d * d; c_sc += 1.f; }
      }
    }
    if (any_bb || any_sc) { acc[0] += s_bb + s_sc; acc[1] += c_bb + c_sc; }
    if (any_bb) { acc[2] += s_bb; acc[3] += c_bb; }
    if (any_sc) { acc[4] += s_sc; acc[5] += c_sc; }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) acc[k] = wave_sum_d(acc[k]);
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 6; ++k) s_red[threadIdx.x >> 6][k] = acc[k];
  __syncthreads();
  if (threadIdx.x < 6) {
    double v = 0;
    for (int w = 0; w < MB / 64; ++w) v += s_red[w][threadIdx.x];
    part[(size_t)blockIdx.x * 6 + threadIdx.x] = v;
  }
}
// pass 2: the partial rows in a fixed order
__global__ void mse_angles_final_kernel(const double *__restrict__ part, float *__restrict__ out) {
  if (threadIdx.x < 6) {
    double v = 0;
    for (int b = 0; b < NBLK; ++b) v += part[(size_t)b * 6 + threadIdx.x];
    out[threadIdx.x] = (float)v;
  }
}

__global__ void mse_angles_bwd_kernel(const float *__restrict__ pred, const float *__restrict__ truth, int64_t T,
                                      const float *__restrict__ sums, float coef, int accumulate,
                                      float *__restrict__ dpred) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const float scale = coef * 2.f / sums[1];
  const float *tp = truth + t * 24, *pp = pred + t * 24;
  float *dp = dpred + t * 24;
  bool any = false;
  for (int k = 0; k < 24; ++k) any |= (tp[k] != 0.f);
  for (int k = 0; k < 24; ++k) {
    const float tv = tp[k];
    const float g = (any && !isnan(tv)) ? scale * (pp[k] - tv) : 0.f;
    dp[k] = accumulate ? dp[k] + g : g;
  }
}

}  // namespace

extern "C" {

size_t ptamd_mse_angles_workspace_bytes(void) { return (size_t)NBLK * 6 * sizeof(double); }

int ptamd_mse_angles_fwd(const float *pred, const float *truth, int64_t T, float *out, void *workspace,
                         size_t workspace_bytes, void *stream) {
  if (T <= 0) return PTAMD_ERR_BAD_SHAPE;
  if (!pt_aligned16(pred) || !pt_aligned16(truth)) return PTAMD_ERR_ALIGN;
  if (!workspace || workspace_bytes < ptamd_mse_angles_workspace_bytes()) return PTAMD_ERR_WORKSPACE;
  double *part = static_cast<double *>(workspace);
  hipLaunchKernelGGL(mse_angles_partial_kernel, dim3(NBLK), dim3(MB), 0, (hipStream_t)stream, pred, truth, T, part);
  int rc = pt_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(mse_angles_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, part, out);
  return pt_check_launch();
}

int ptamd_mse_angles_bwd(const float *pred, const float *truth, int64_t T, const float *sums, float coef,
                         int accumulate, float *dpred, void *stream) {
  if (T <= 0) return PTAMD_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(mse_angles_bwd_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     pred, truth, T, sums, coef, accumulate, dpred);
  return pt_check_launch();
}

}  // extern "C"
