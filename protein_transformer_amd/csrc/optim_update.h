// The per-element update expressions of the optimizers (train.py:41-46,371-381: clip coefficient, SGD / Adam with L2 weight
// decay), shared by the plain kernels of optim.hip and the fused step + weight-preparation kernel of wprep.hip.  Every
// multiply-add is written out (fmaf) and contraction is off inside: the two kernels vectorise differently, and left to the
// compiler the choice of WHICH product of `a * b + c * d` is fused could differ between them - with these expressions both
// paths produce the same bits.
#pragma once
#include "common.h"

namespace ptopt {

__device__ __forceinline__ float clip_coef(const float *sqnorm, float max_norm) {
  if (max_norm <= 0.f || sqnorm == nullptr) return 1.f;
  const float c = max_norm / (sqrtf(sqnorm[0]) + 1e-6f);
  return c < 1.f ? c : 1.f;
}
// p - lr * (coef * d + wd * p)
__device__ __forceinline__ float sgd_update(float p, float d, float coef, float lr, float wd) {
#pragma clang fp contract(off)
  const float t = fmaf(coef, d, wd * p);
  return fmaf(-lr, t, p);
}
// m, v: first / second moment of the element, updated in place
__device__ __forceinline__ float adam_update(float p, float g, float &m, float &v, float coef, float wd, float beta1, float beta2,
                                             float eps, float step_size, float inv_sqrt_bc2) {
#pragma clang fp contract(off)
  const float gr = fmaf(coef, g, wd * p);
  const float mi = fmaf(beta1, m, (1.f - beta1) * gr);
  const float vi = fmaf(beta2, v, ((1.f - beta2) * gr) * gr);
  m = mi;
  v = vi;
  const float denom = fmaf(sqrtf(vi), inv_sqrt_bc2, eps);
  return p - (step_size * mi) / denom;
}

}  // namespace ptopt
