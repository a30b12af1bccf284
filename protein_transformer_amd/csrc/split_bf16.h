// Exact three-term bf16 splitting of f32 values and the LDS tile format of the split-bf16 attention kernels
// (see gemm_split_kernel.h for the arithmetic: x = t1 + t2 + t3 exactly, products evaluated as six bf16 MFMAs).
#pragma once
#include "common.h"

namespace ptsplit {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));  // v_cvt_pk_bf16_f32, round to nearest even
}
// (x0, x1) -> three packed bf16 pairs with x = t1 + t2 + t3 exactly.  The residuals are taken with SCALAR v_sub_f32:
// left alone the compiler pairs them into v_pk_add_f32, and packed f32 VALU instructions stall the matrix pipe
// (MI355X_MICROARCH.md, "price of one filler beside MFMAs"); the empty asm statements keep the subtractions apart.
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t &t1, uint32_t &t2, uint32_t &t3) {
  t1 = pack_bf16(x0, x1);
  float r0 = x0 - __uint_as_float(t1 << 16);
  asm volatile("" : "+v"(r0));
  float r1 = x1 - __uint_as_float(t1 & 0xffff0000u);
  asm volatile("" : "+v"(r1));
  t2 = pack_bf16(r0, r1);
  float q0 = r0 - __uint_as_float(t2 << 16);
  asm volatile("" : "+v"(q0));
  float q1 = r1 - __uint_as_float(t2 & 0xffff0000u);
  asm volatile("" : "+v"(q1));
  t3 = pack_bf16(q0, q1);
}

// Two f16 terms of the SCALED value: x s = h1 + h2 + e with |e| <= 2^-22 |x s| (2^-24 absolute where h2 is subnormal).
// s is a power of two that takes the largest |x| of the operand row into [2^14, 2^15), so x s is exact, h1 cannot
// overflow and the residual x s - h1 (one fma, exact) stays a normal f16 for values down to 2^-18 of the row maximum.
// Scalar operations for the reason given above (v_pk_mul_f32 / v_pk_fma_f32 beside MFMAs).
__device__ __forceinline__ void split_pair_f16(float x0, float x1, float s0, float s1, uint32_t &t1, uint32_t &t2) {
  float a0 = x0 * s0;
  asm volatile("" : "+v"(a0));  // (between the two products: with two DIFFERENT scales the compiler pairs them into a v_pk_mul_f32 otherwise)
  float a1 = x1 * s1;
  asm volatile("" : "+v"(a1));
  const f16x2 h = __builtin_convertvector((f32x2){a0, a1}, f16x2);  // v_cvt_pk_f16_f32, round to nearest even
  float r0 = fmaf(x0, s0, -(float)h[0]), r1 = fmaf(x1, s1, -(float)h[1]);  // v_fma_mix_f32
  asm volatile("" : "+v"(r0));
  asm volatile("" : "+v"(r1));
  const f16x2 g = __builtin_convertvector((f32x2){r0, r1}, f16x2);
  t1 = __builtin_bit_cast(uint32_t, h);
  t2 = __builtin_bit_cast(uint32_t, g);
}
// Four values at once, each with its own scale.  PT_MIX_SPLIT (ablation, slower: profiles/r02/r02_f16_split_mix_ablation.txt):
// EIGHT vector instructions instead of twelve - v_fma_mixlo/mixhi_f16 round the f32 product x s straight into one half of
// a packed register and take the f16 half back as the addend of the residual fma(x, s, -h) - same results, bit for bit.
// The two pairs are interleaved so that no instruction reads a half-register write of the instruction before it
// (gfx940+ destination-select forwarding hazard, which the compiler cannot see inside an asm block), and one s_nop
// separates the last write from whatever the compiler schedules next.
__device__ __forceinline__ void split_quad_f16(float x0, float x1, float x2, float x3, float s0, float s1, float s2,
                                               float s3, uint2 &t1, uint2 &t2) {
#ifndef PT_MIX_SPLIT
  // The four products one by one, each behind the one before it: with four DIFFERENT scales (row-contiguous operands: a
  // float4 is four rows) the compiler otherwise pairs them into v_pk_mul_f32 - 24 per stage in the producers of the
  // weight-gradient products, each contending with the consumers' MFMAs for the pipe (profiles/r03/r03_gemm_stage_trace.txt).
  float a0 = x0 * s0;
  asm volatile("" : "+v"(a0), "+v"(x1));
  float a1 = x1 * s1;
  asm volatile("" : "+v"(a1), "+v"(x2));
  float a2 = x2 * s2;
  asm volatile("" : "+v"(a2), "+v"(x3));
  float a3 = x3 * s3;
  asm volatile("" : "+v"(a3));
  const f16x2 h01 = __builtin_convertvector((f32x2){a0, a1}, f16x2), h23 = __builtin_convertvector((f32x2){a2, a3}, f16x2);
  float r0 = fmaf(x0, s0, -(float)h01[0]), r1 = fmaf(x1, s1, -(float)h01[1]);  // v_fma_mix_f32
  float r2 = fmaf(x2, s2, -(float)h23[0]), r3 = fmaf(x3, s3, -(float)h23[1]);
  asm volatile("" : "+v"(r0));
  asm volatile("" : "+v"(r1));
  asm volatile("" : "+v"(r2));
  asm volatile("" : "+v"(r3));
  const f16x2 g01 = __builtin_convertvector((f32x2){r0, r1}, f16x2), g23 = __builtin_convertvector((f32x2){r2, r3}, f16x2);
  t1 = make_uint2(__builtin_bit_cast(uint32_t, h01), __builtin_bit_cast(uint32_t, h23));
  t2 = make_uint2(__builtin_bit_cast(uint32_t, g01), __builtin_bit_cast(uint32_t, g23));
#else
  uint32_t ha, hb, ga, gb;
  asm("v_fma_mixlo_f16 %0, %4, %8, 0\n\t"
      "v_fma_mixhi_f16 %0, %5, %9, 0\n\t"
      "v_fma_mixlo_f16 %1, %6, %10, 0\n\t"
      "v_fma_mixhi_f16 %1, %7, %11, 0\n\t"
      "v_fma_mixlo_f16 %2, %4, %8, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %2, %5, %9, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %3, %6, %10, -%1 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %3, %7, %11, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "s_nop 0"
      : "=&v"(ha), "=&v"(hb), "=&v"(ga), "=&v"(gb)
      : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(s0), "v"(s1), "v"(s2), "v"(s3));
  t1 = make_uint2(ha, hb);
  t2 = make_uint2(ga, gb);
#endif
}
// eight f32 -> the three bf16x8 MFMA operands
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 (&f)[3]) {
  uint32_t t[3][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split_pair(x[2 * i], x[2 * i + 1], t[0][i], t[1][i], t[2][i]);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const u32x4 u = {t[k][0], t[k][1], t[k][2], t[k][3]};  // (a register vector, so the packs land in place)
    f[k] = __builtin_bit_cast(bf16x8, u);
  }
}

// six-product f32-grade multiply-accumulate of split operands, smallest products first
__device__ __forceinline__ f32x16 mfma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
  return c;
}

// ---- LDS image of a [ROWS][64] f32 tile as three bf16 planes.  A row is 96 bf16 (192 B: 128 B of data + pad); the
// 16-byte chunk c = d / 8 of row r is stored at chunk c ^ ((r >> 2) & 3).  With this stride + swizzle BOTH fragment
// reads are bank-conflict free:
//   frag_rows : rows of the tile are MFMA rows, k runs along d   (one ds_read_b128 per plane)
//   frag_cols : columns (d) of the tile are MFMA rows, k runs along the tile rows (two ds_read_b64_tr_b16 per plane)
constexpr int T64_LD = 96;
template <int ROWS>
struct Tile64 {
  static constexpr int PLANE = ROWS * T64_LD;   // bf16 elements per plane
  static constexpr int ELEMS = 3 * PLANE;
  static __device__ __forceinline__ int offset(int row, int d) {  // element offset of (row, d), d multiple of 4
    return row * T64_LD + ((((d >> 3) ^ (row >> 2)) & 3) | ((d >> 3) & 4)) * 8 + (d & 7);
  }
  // one float4 = 4 consecutive d of one row
  static __device__ __forceinline__ void store4(unsigned short *__restrict__ s, int row, int d, const float4 &v) {
    uint2 t1, t2, t3;
    split_pair(v.x, v.y, t1.x, t2.x, t3.x);
    split_pair(v.z, v.w, t1.y, t2.y, t3.y);
    const int off = offset(row, d);
    *reinterpret_cast<uint2 *>(s + off) = t1;
    *reinterpret_cast<uint2 *>(s + PLANE + off) = t2;
    *reinterpret_cast<uint2 *>(s + 2 * PLANE + off) = t3;
  }
  // MFMA operand: lane l holds tile row r0 + (l & 31), d = 16 step + 8 (l >> 5) + 0..7
  static __device__ __forceinline__ void frag_rows(const unsigned short *__restrict__ s, int r0, int step, int lane,
                                                   bf16x8 (&f)[3]) {
    const unsigned short *q = s + offset(r0 + (lane & 31), 16 * step + 8 * (lane >> 5));
#pragma unroll
    for (int t = 0; t < 3; ++t) f[t] = *reinterpret_cast<const bf16x8 *>(q + t * PLANE);
  }
  // MFMA operand: lane l holds column d0 + (l & 31) of the tile rows kb + 4 (l >> 5) + {0..3} (elements 0..3) and
  // kb + 8 + 4 (l >> 5) + {0..3} (elements 4..7); kb multiple of 16.  This is the k order in which a 32x32 MFMA
  // accumulator holds its rows (register r, lane half h -> row (r & 3) + 8 (r >> 2) + 4 h), so an accumulator can be
  // fed back as the other operand without any lane exchange.
  static __device__ __forceinline__ void frag_cols(const unsigned short *__restrict__ s, int kb, int d0, int lane,
                                                   bf16x8 (&f)[3]) {
    const int q16 = lane & 15;
    const int row = kb + 4 * (lane >> 5) + (q16 >> 2);
    const unsigned short *q0 = s + offset(row, d0 + (lane & 16) + 4 * (q16 & 3));
    const unsigned short *q1 = s + offset(row + 8, d0 + (lane & 16) + 4 * (q16 & 3));
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(q0 + t * PLANE));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(q1 + t * PLANE));
      const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      f[t] = __builtin_bit_cast(bf16x8, both);
    }
  }
};

}  // namespace ptsplit
