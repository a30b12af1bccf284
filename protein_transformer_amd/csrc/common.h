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

// ---- counter-based random words for dropout.  Masks are regenerated in the backward pass instead of stored, so the
// generator sits in GEMM epilogues and attention inner loops: integer multiplies run at a quarter of the VALU rate on
// CDNA (a Philox4x32-10 call costs ~900 cycles per wavefront), so the mixer is built from full-rate shifts, adds and
// xors only: Thomas Wang's 32-bit integer hash with its multiplication by 2057 written as shifts.  Uniformity and
// independence of the masks it produces are checked in tests/test_host_logic.py (numpy restatement) and the device
// output is compared with that restatement in tests/test_gpu_kernels.py.
__device__ __forceinline__ uint32_t pt_mix32(uint32_t x) {
  x = ~x + (x << 15);
  x ^= x >> 12;
  x += x << 2;
  x ^= x >> 4;
  x = x + (x << 3) + (x << 11);
  x ^= x >> 16;
  return x;
}
// four uniform 32-bit words for counter `idx` under the key (seed, stream_id)
__device__ __forceinline__ uint4 pt_rand4(uint64_t seed, uint64_t idx, uint32_t stream_id) {
  uint32_t h = pt_mix32((uint32_t)idx ^ (uint32_t)seed);
  h = pt_mix32(h ^ (uint32_t)(idx >> 32) ^ (uint32_t)(seed >> 32) ^ (stream_id * 0x9E3779B9u));
  return make_uint4(pt_mix32(h ^ 0x68E31DA4u), pt_mix32(h ^ 0xB5297A4Du), pt_mix32(h ^ 0x1B56C4E9u),
                    pt_mix32(h ^ 0x7F4A7C15u));
}
// ---- dropout mask of a [rows, cols] activation (GEMM epilogues, ptamd_dropout_bwd).  One pt_rand4 call serves EIGHT
// rows of a column - the 16-bit halves of its four words against a 16-bit threshold - namely the rows that one lane of a
// 32 x 32 MFMA accumulator tile holds in two of its four register groups: row = 32 I + 8 g + 4 h + e (g = 0..3,
// h = lane half, e = 0..3) belongs to call (I, h, g >> 1) and field (g & 1) * 4 + e of that call.
__device__ __forceinline__ uint64_t drop_call_index(int64_t row, int col, int cols) {
  const int64_t call_row = ((row >> 5) << 2) | (((row >> 2) & 1) << 1) | ((row >> 4) & 1);
  return (uint64_t)call_row * (uint64_t)cols + (uint64_t)col;
}
__device__ __forceinline__ int drop_field(int64_t row) { return (int)(((row >> 3) & 1) * 4 + (row & 3)); }
__device__ __forceinline__ uint32_t drop_field_value(const uint4 &r, int f) {
  const uint32_t w = (f >> 1) == 0 ? r.x : (f >> 1) == 1 ? r.y : (f >> 1) == 2 ? r.z : r.w;
  return (f & 1) ? w >> 16 : w & 0xffffu;
}
// one element (slow paths: scalar epilogues, split-K reduce); thr16 = dropout_threshold(p) >> 16
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint32_t stream_id, int64_t row, int col, int cols, uint32_t thr16) {
  return drop_field_value(pt_rand4(seed, drop_call_index(row, col, cols), stream_id), drop_field(row)) >= thr16;
}
// f16x2 arithmetic (include/ptamd.h, PTAMD_GEMM_F16X2): scale (bits of a power of two) of an operand row whose largest
// |x| - or an upper bound of it - has the bits `amax`: amax * scale in [2^14, 2^15); rows of zeros get the largest
// finite power.  A larger maximum gives a SMALLER scale.
__device__ __forceinline__ uint32_t pt_row_scale_bits(uint32_t amax) { return min(268u - (amax >> 23), 254u) << 23; }

// keep-threshold on the 32-bit word: keep iff word >= p * 2^32
__device__ __forceinline__ uint32_t dropout_threshold(float p) {
  double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}
