// Dropout on the attention probabilities (Attention.py:16-22 of the reference: nn.Dropout(0.1) on softmax(QK^T)),
// shared by attention.hip and attention_split.hip so that every kernel of both arithmetic modes draws the same mask.
//
// One 32-bit word of the counter hash pt_mix32 (common.h) serves a PAIR of adjacent keys of one query: the even key
// is decided by its low 16 bits, the odd key by its high 16 bits, against a 16-bit threshold.  The kernels that keep a
// query per lane hold such pairs in adjacent accumulator registers, so they evaluate the hash once per two elements.
// The drop probability is therefore quantised to thr16 / 65536 (0.1 -> 0.100006) and the kept probabilities are
// scaled by exactly 65536 / (65536 - thr16), so the expectation is preserved.  Masks are regenerated in the backward
// kernels instead of being stored - except that the f16x2 forward kernel can hand its decisions (1 bit each) to the fused
// backward kernel, which then skips the generator (attn_drop_keys_in_rows_export below); same decisions either way.
#pragma once
#include "common.h"

struct AttnDrop {
  uint32_t lo, hi, thr16;
  float ks;
};
constexpr uint32_t ATTN_C_Q = 0x9E3779B1u, ATTN_C_K = 0x85EBCA77u;

__device__ __forceinline__ AttnDrop make_attn_drop(uint64_t seed, uint32_t stream_id, uint32_t bh, float p) {
  AttnDrop d;
  d.lo = (uint32_t)seed ^ (bh * 0xC2B2AE35u);
  d.hi = (uint32_t)(seed >> 32) ^ (stream_id * 0x27D4EB2Fu) ^ bh;
  const float t = p * 65536.f + 0.5f;
  d.thr16 = t >= 65535.f ? 65535u : (uint32_t)t;
  d.ks = 65536.f / (65536.f - (float)d.thr16);
  return d;
}
// the word of (query q, key pair kp = key >> 1), from its two precomputed halves
__device__ __forceinline__ uint32_t attn_q_part(const AttnDrop &d, uint32_t q) { return q * ATTN_C_Q + d.lo; }
__device__ __forceinline__ uint32_t attn_kp_part(const AttnDrop &d, uint32_t kp) { return kp * ATTN_C_K + d.hi; }
__device__ __forceinline__ uint32_t attn_word(uint32_t q_part, uint32_t kp_part) { return pt_mix32(q_part ^ kp_part); }
__device__ __forceinline__ bool attn_keep_even(const AttnDrop &d, uint32_t w) { return (w & 0xffffu) >= d.thr16; }
__device__ __forceinline__ bool attn_keep_odd(const AttnDrop &d, uint32_t w) { return (w >> 16) >= d.thr16; }

// Accumulator layout of a 32 x 32 MFMA tile: register r of lane half lh holds row (r & 3) + 8 (r >> 2) + 4 lh.
//
// Queries in lanes, keys in rows (forward and dQ kernels), k0 = first key of the 32-key block (even): registers 2j and
// 2j + 1 hold the even and the odd key of pair (k0 >> 1) + 2 lh + (j & 1) + 4 (j >> 1).  Returns bit r = keep.
__device__ __forceinline__ uint32_t attn_keep_bits_keys_in_rows(const AttnDrop &d, uint32_t q_part, int k0, int lh) {
  const uint32_t base = attn_kp_part(d, (uint32_t)((k0 >> 1) + 2 * lh));
  uint32_t bits = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t w = attn_word(q_part, base + (uint32_t)((j & 1) + 4 * (j >> 1)) * ATTN_C_K);
    bits |= (attn_keep_even(d, w) ? 1u : 0u) << (2 * j);
    bits |= (attn_keep_odd(d, w) ? 2u : 0u) << (2 * j);
  }
  return bits;
}
// The same decisions APPLIED to the 16 accumulator values of a lane instead of collected as bits: one 16-bit compare on a
// half of the word and one select per element (the bit mask costs a compare, a select, a shift and an or per element to
// build and two more to apply).  x[2j], x[2j + 1] belong to the word of pair j as above.
template <typename V>
__device__ __forceinline__ void attn_drop_keys_in_rows(const AttnDrop &d, uint32_t q_part, int k0, int lh, V &x) {
  const uint32_t base = attn_kp_part(d, (uint32_t)((k0 >> 1) + 2 * lh));
  const uint16_t t16 = (uint16_t)d.thr16;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t w = attn_word(q_part, base + (uint32_t)((j & 1) + 4 * (j >> 1)) * ATTN_C_K);
    x[2 * j] = (uint16_t)w >= t16 ? x[2 * j] : 0.f;
    x[2 * j + 1] = (uint16_t)(w >> 16) >= t16 ? x[2 * j + 1] : 0.f;
  }
}
// The same, and the decisions EXPORTED for a backward kernel that keeps keys in lanes (attn_bwd_fused_f16x2_kernel reads them
// instead of drawing the words again: the generator is 29 % of that kernel's VALU instructions).  The compare of an element
// already leaves its 64 lane decisions in a scalar register pair - lanes 0..31: the 32 queries against the key of lane half
// 0, lanes 32..63: against the key four rows further - so the export is two v_writelane per compare: the returned word of
// lane i < 32 holds, in bit q, whether query (lane q) keeps key k0 + i.
// v_writelane_b32 through the LLVM intrinsic (this clang has no builtin for it; as inline asm the compiler does not see the
// wait states the instruction needs behind the v_cmp that wrote its scalar source - measured: two stale words per tile)
extern "C" __device__ int pt_llvm_amdgcn_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane");
template <int LANE>
__device__ __forceinline__ void attn_writelane(int &word, uint32_t value) {
  word = pt_llvm_amdgcn_writelane((int)value, LANE, word);
}
template <int J, typename V>
__device__ __forceinline__ void attn_drop_export_pair(const AttnDrop &d, uint32_t q_part, uint32_t base, uint16_t t16, V &x, int &word) {
  const uint32_t w = attn_word(q_part, base + (uint32_t)((J & 1) + 4 * (J >> 1)) * ATTN_C_K);
  const bool ke = (uint16_t)w >= t16, ko = (uint16_t)(w >> 16) >= t16;
  x[2 * J] = ke ? x[2 * J] : 0.f;
  x[2 * J + 1] = ko ? x[2 * J + 1] : 0.f;
  const uint64_t me = __builtin_amdgcn_ballot_w64(ke), mo = __builtin_amdgcn_ballot_w64(ko);
  constexpr int OFF = 2 * (J & 1) + 8 * (J >> 1);   // row of register 2 J in lane half 0 (lane half 1: + 4)
  attn_writelane<OFF>(word, (uint32_t)me);
  attn_writelane<OFF + 4>(word, (uint32_t)(me >> 32));
  attn_writelane<OFF + 1>(word, (uint32_t)mo);
  attn_writelane<OFF + 5>(word, (uint32_t)(mo >> 32));
}
template <typename V>
__device__ __forceinline__ uint32_t attn_drop_keys_in_rows_export(const AttnDrop &d, uint32_t q_part, int k0, int lh, V &x) {
  const uint32_t base = attn_kp_part(d, (uint32_t)((k0 >> 1) + 2 * lh));
  const uint16_t t16 = (uint16_t)d.thr16;
  int word = 0;
  attn_drop_export_pair<0>(d, q_part, base, t16, x, word);
  attn_drop_export_pair<1>(d, q_part, base, t16, x, word);
  attn_drop_export_pair<2>(d, q_part, base, t16, x, word);
  attn_drop_export_pair<3>(d, q_part, base, t16, x, word);
  attn_drop_export_pair<4>(d, q_part, base, t16, x, word);
  attn_drop_export_pair<5>(d, q_part, base, t16, x, word);
  attn_drop_export_pair<6>(d, q_part, base, t16, x, word);
  attn_drop_export_pair<7>(d, q_part, base, t16, x, word);
  return (uint32_t)word;
}
// layout of the exported decisions: word [(protein, head)][query tile of 32][key], keys padded to a multiple of 32
__host__ __device__ inline size_t attn_keep_words(int B, int L, int H) {
  const size_t nt = (size_t)(L + 31) / 32;
  return (size_t)B * H * nt * (nt * 32);
}
// Keys in lanes, queries in rows (dK/dV kernel), q0 = first query of the 32-query block: register r holds query
// q0 + 4 lh + (r & 3) + 8 (r >> 2); the lane's key selects the half of every word.  Returns bit r = keep.
__device__ __forceinline__ uint32_t attn_keep_bits_queries_in_rows(const AttnDrop &d, uint32_t key, int q0, int lh) {
  const uint32_t kp_part = attn_kp_part(d, key >> 1), shift = (key & 1u) * 16u;
  const uint32_t base = attn_q_part(d, (uint32_t)(q0 + 4 * lh));
  uint32_t bits = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const uint32_t w = attn_word(base + (uint32_t)((r & 3) + 8 * (r >> 2)) * ATTN_C_Q, kp_part);
    bits |= (((w >> shift) & 0xffffu) >= d.thr16 ? 1u : 0u) << r;
  }
  return bits;
}

// The same bits at half the generator calls, for kernels whose lanes 2k and 2k + 1 hold the keys 2m and 2m + 1 of one
// pair (even first key of the wavefront, same q0 and lh in both lanes): the two lanes need the SAME 16 words, one its
// low and the other its high halves.  The even lane draws the words of registers 0..7, the odd lane those of 8..15, both
// decide both halves, and one DPP exchange between neighbours hands over the eight decisions the partner needs.
__device__ __forceinline__ uint32_t attn_keep_bits_queries_in_rows_paired(const AttnDrop &d, uint32_t key, int q0, int lh) {
  const uint32_t kp_part = attn_kp_part(d, key >> 1), odd = key & 1u;
  const uint32_t base = attn_q_part(d, (uint32_t)(q0 + 4 * lh) + 16u * odd);
  uint32_t lo_bits = 0, hi_bits = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t w = attn_word(base + (uint32_t)((i & 3) + 8 * (i >> 2)) * ATTN_C_Q, kp_part);
    lo_bits |= ((w & 0xffffu) >= d.thr16 ? 1u : 0u) << i;
    hi_bits |= ((w >> 16) >= d.thr16 ? 1u : 0u) << i;
  }
  const uint32_t mine = odd ? hi_bits : lo_bits, give = odd ? lo_bits : hi_bits;
  const uint32_t got = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  return odd ? (got | (mine << 8)) : (mine | (got << 8));
}
