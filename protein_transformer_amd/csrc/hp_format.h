// The "half-pair" (hp) operand format of the pre-split GEMM (gemm_hp.hip): an fp32 matrix stored as TWO f16 planes plus one
// power-of-two scale per row - the f16x2 arithmetic of include/ptamd.h with the splitting done ONCE, by whoever writes the
// operand, instead of by every GEMM that reads it.
//
//   x[r][c] * scale[r] = hi[r][c] + lo[r][c] + e,   |e| <= 2^-22 |x scale|     (2^-25 absolute where lo is subnormal)
//   scale[r] = 2^e with  max_c |x[r][c]| * scale[r]  in [2^14, 2^15)   (or any smaller power of two: a writer that only
//   knows an upper bound of the row maximum uses the bound - every binade of slack costs one binade of the 18-binade
//   window in which an element keeps its full 22 bits, nothing else).
//
// Memory layout (4 bytes per element, like the fp32 it replaces): the matrix is cut into BLOCKS of 32 rows x 16 columns;
// block (rb, kb) of plane p is the 1 KiB at byte offset (((rb * KB16 + kb) * 2) + p) * 1024 with KB16 = Kp / 16,
// Kp = K rounded up to 32 (zero filled), rows rounded up to 32 (zero filled, scale 1).  Inside a block the 16-byte chunk
// that holds columns 8 h .. 8 h + 7 (h = 0, 1) of row r sits at chunk index  2 r + (h ^ ((r >> 3) & 1)).
//
// Why this shape: a block is exactly the image ONE `global_load_lds_dwordx4` wave instruction writes into LDS (64 lanes x
// 16 B, lane-linear), fully coalesced on the global side, and exactly what ONE `ds_read_b128` needs as the A / B operand
// of v_mfma_f32_32x32x16_f16 (lane l: row l & 31, k = 8 (l >> 5) .. +7).  The XOR puts the 16 lanes of every
// ds_read_b128 service group on 16 different 16-byte slots of the 256-byte bank row: conflict-free without padding (a
// padded image cannot be written by LDS-DMA).  So a GEMM stage costs no VALU, no VGPR and no ds_write.
#pragma once
#include "common.h"

namespace pthp {

constexpr int BLK_ROWS = 32, BLK_COLS = 16, BLK_BYTES = 1024;

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
__host__ __device__ inline int kb16(int K) { return round_up(K, 32) / 16; }
__host__ __device__ inline size_t plane_bytes(int rows, int K) {
  return (size_t)(round_up(rows, 32) / 32) * kb16(K) * 2 * BLK_BYTES;
}
// byte offset of block (rb, kb) of plane p
__host__ __device__ inline size_t block_offset(int rb, int kb, int p, int KB16) {
  return ((size_t)((size_t)rb * KB16 + kb) * 2 + p) * BLK_BYTES;
}
// 16-byte chunk index (0..63) of (row r in 0..31, column half h in 0..1) inside a block
__host__ __device__ inline int chunk_index(int r, int h) { return 2 * r + (h ^ ((r >> 3) & 1)); }
// inverse: chunk index -> (r, h)
__host__ __device__ inline void chunk_coords(int c, int &r, int &h) {
  r = c >> 1;
  h = (c & 1) ^ ((r >> 3) & 1);
}

// scale (bits of a power of two) for a row whose largest |x| (or an upper bound of it) has the bits `amax`:
// amax * scale in [2^14, 2^15); rows of zeros / subnormals get the largest finite power (their planes are zero anyway)
__device__ __forceinline__ uint32_t scale_bits_of(uint32_t amax) { return min(268u - (amax >> 23), 254u) << 23; }
__device__ __forceinline__ float inverse_of_scale(float scale) { return __uint_as_float((254u << 23) - __float_as_uint(scale)); }

#ifdef __HIPCC__
// Four consecutive columns c .. c + 3 (c multiple of 4) of row `row`, already multiplied by nothing: x * s is split here
// into the two planes (8 bytes each).  Writers that hold a row in float4 pieces (LayerNorm, the weight splitter) use this.
__device__ __forceinline__ void store4_split(char *__restrict__ planes, int KB16, int64_t row, int c, float4 v, float s) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const float a0 = v.x * s, a1 = v.y * s, a2 = v.z * s, a3 = v.w * s;               // exact: s is a power of two
  const h2 h01 = __builtin_convertvector((f2){a0, a1}, h2), h23 = __builtin_convertvector((f2){a2, a3}, h2);
  const h2 l01 = __builtin_convertvector((f2){a0 - (float)h01[0], a1 - (float)h01[1]}, h2);
  const h2 l23 = __builtin_convertvector((f2){a2 - (float)h23[0], a3 - (float)h23[1]}, h2);
  const int r = (int)(row & 31), h = (c >> 3) & 1;
  char *dst = planes + block_offset((int)(row >> 5), c >> 4, 0, KB16) + chunk_index(r, h) * 16 + ((c >> 2) & 1) * 8;
  *reinterpret_cast<uint2 *>(dst) = make_uint2(__builtin_bit_cast(uint32_t, h01), __builtin_bit_cast(uint32_t, h23));
  *reinterpret_cast<uint2 *>(dst + BLK_BYTES) = make_uint2(__builtin_bit_cast(uint32_t, l01), __builtin_bit_cast(uint32_t, l23));
}
#endif

}  // namespace pthp
