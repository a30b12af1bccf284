// Shared helpers for the gfx950 kernels of libptamd (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ptamd.h"

#define PT_WAVE 64

extern thread_local hipError_t g_pt_last_hip_error;

static inline int pt_check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_pt_last_hip_error = e;
    return PTAMD_ERR_HIP;
  }
  return PTAMD_OK;
}

#define PT_HIP_TRY(expr)                 \
  do {                                   \
    hipError_t _e = (expr);              \
    if (_e != hipSuccess) {              \
      g_pt_last_hip_error = _e;          \
      return PTAMD_ERR_HIP;              \
    }                                    \
  } while (0)

static inline bool pt_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- wave64 reductions (DPP/bpermute via __shfl_xor; all 64 lanes must participate)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- counter-based RNG for dropout: Philox4x32-10 keyed by (seed), counter = (index, stream).
// One call yields 4 uniform 32-bit words; masks are regenerated in the backward pass instead of stored.
__device__ __forceinline__ uint4 philox4x32(uint64_t seed, uint64_t idx, uint32_t stream_id) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), c2 = stream_id, c3 = 0x9E3779B9u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}
// keep-threshold on the 32-bit word: keep iff word >= p * 2^32
__device__ __forceinline__ uint32_t dropout_threshold(float p) {
  double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}
