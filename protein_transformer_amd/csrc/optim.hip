// Gradient clipping and the optimizer step over ONE flat fp32 parameter buffer (HBM-bound, 16-byte accesses).
//
// Replaces, for the whole model in three launches,
//   torch.nn.utils.clip_grad_norm_(model.parameters(), args.clip)   /root/reference/protein_transformer/train.py:41-43
//   optimizer.step() with SGD(lr, weight_decay=0.01) or Adam(betas=(0.9,0.98), eps=1e-9, weight_decay=0.01)
//                                                                   .../train.py:46,371-381
// The clip coefficient min(1, max_norm / (||g|| + 1e-6)) is computed on the device from the squared norm, so the
// step never synchronises with the host.  Weight decay is the L2 form torch.optim uses (g + wd * w).
#include "common.h"
#include "optim_update.h"

namespace {
using ptopt::clip_coef;

constexpr int SQ_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float *__restrict__ g, int64_t n,
                                                             double *__restrict__ part) {
  __shared__ double s_red[4];
  const int64_t n4 = n >> 2;
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)SQ_BLOCKS * 256) {
    const float4 v = reinterpret_cast<const float4 *>(g)[i];
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = g[(n4 << 2) + threadIdx.x];
    s += v * v;
  }
  double d = wave_sum_d((double)s);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}
__global__ __launch_bounds__(256) void sqnorm_final_kernel(const double *__restrict__ part, float *__restrict__ out) {
  __shared__ double s_red[4];
  double d = 0;
  for (int i = threadIdx.x; i < SQ_BLOCKS; i += 256) d += part[i];
  d = wave_sum_d(d);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
}

// ZERO: the gradient is zeroed behind its last read - the `optimizer.zero_grad()` of the NEXT step (train.py:37) folded into
// the pass that streams g anyway (a 76 MB fill of its own otherwise)
template <bool ZERO>
__global__ void sgd_kernel(float *__restrict__ w, float *__restrict__ g, int64_t n, const float *sqnorm,
                           float max_norm, float lr, float wd) {
  const float coef = clip_coef(sqnorm, max_norm);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n4 = n >> 2;
  if (i < n4) {
    float4 p = reinterpret_cast<float4 *>(w)[i];
    const float4 d = reinterpret_cast<const float4 *>(g)[i];
    p.x = ptopt::sgd_update(p.x, d.x, coef, lr, wd);
    p.y = ptopt::sgd_update(p.y, d.y, coef, lr, wd);
    p.z = ptopt::sgd_update(p.z, d.z, coef, lr, wd);
    p.w = ptopt::sgd_update(p.w, d.w, coef, lr, wd);
    reinterpret_cast<float4 *>(w)[i] = p;
    if (ZERO) reinterpret_cast<float4 *>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  } else if (i < n4 + (n & 3)) {
    const int64_t k = (n4 << 2) + (i - n4);
    w[k] = ptopt::sgd_update(w[k], g[k], coef, lr, wd);
    if (ZERO) g[k] = 0.f;
  }
}

template <bool ZERO>
__global__ void adam_kernel(float *__restrict__ w, float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, int64_t n, const float *sqnorm, float max_norm, float step_size,
                            float beta1, float beta2, float eps, float wd, float inv_sqrt_bc2) {
  const float coef = clip_coef(sqnorm, max_norm);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float mi = m[i], vi = v[i];
  w[i] = ptopt::adam_update(w[i], g[i], mi, vi, coef, wd, beta1, beta2, eps, step_size, inv_sqrt_bc2);
  m[i] = mi;
  v[i] = vi;
  if (ZERO) g[i] = 0.f;
}

}  // namespace

extern "C" {

size_t ptamd_grad_sqnorm_workspace_bytes(void) { return SQ_BLOCKS * sizeof(double); }

int ptamd_grad_sqnorm(const float *g, int64_t n, float *out, void *workspace, size_t workspace_bytes, void *stream) {
  if (n <= 0) return PTAMD_ERR_BAD_SHAPE;
  if (!workspace || workspace_bytes < SQ_BLOCKS * sizeof(double)) return PTAMD_ERR_WORKSPACE;
  if (!pt_aligned16(g)) return PTAMD_ERR_ALIGN;
  double *part = static_cast<double *>(workspace);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(SQ_BLOCKS), dim3(256), 0, st, g, n, part);
  int rc = pt_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, st, part, out);
  return pt_check_launch();
}

int ptamd_sgd_step(float *w, float *g, int64_t n, const float *sqnorm, float max_norm, float lr,
                   float weight_decay, int zero_grad, void *stream) {
  if (n <= 0) return PTAMD_ERR_BAD_SHAPE;
  if (!pt_aligned16(w) || !pt_aligned16(g)) return PTAMD_ERR_ALIGN;
  const int64_t items = (n >> 2) + (n & 3);
  const dim3 grid((unsigned)((items + 255) / 256));
  if (zero_grad) hipLaunchKernelGGL(sgd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, w, g, n, sqnorm, max_norm, lr, weight_decay);
  else hipLaunchKernelGGL(sgd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, w, g, n, sqnorm, max_norm, lr, weight_decay);
  return pt_check_launch();
}

int ptamd_adam_step(float *w, float *g, float *m, float *v, int64_t n, const float *sqnorm, float max_norm,
                    float lr, float beta1, float beta2, float eps, float weight_decay, int step, int zero_grad, void *stream) {
  if (n <= 0 || step <= 0) return PTAMD_ERR_BAD_SHAPE;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const dim3 grid((unsigned)((n + 255) / 256));
  const float ss = (float)(lr / bc1), ib = (float)(1.0 / sqrt(bc2));
  if (zero_grad) hipLaunchKernelGGL(adam_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, w, g, m, v, n, sqnorm, max_norm, ss, beta1, beta2, eps, weight_decay, ib);
  else hipLaunchKernelGGL(adam_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, w, g, m, v, n, sqnorm, max_norm, ss, beta1, beta2, eps, weight_decay, ib);
  return pt_check_launch();
}

}  // extern "C"
