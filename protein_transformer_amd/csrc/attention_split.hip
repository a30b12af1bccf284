// Fused masked multi-head self-attention on the bf16 matrix pipe with exact three-term operand splitting (dk = 64 and 32).
//
// Same operator, interface, masking, dropout hash and outputs as attention.hip
//   /root/reference/protein_transformer/models/transformer/Attention.py:14-22,55-68
// but every f32 matrix product (QK^T, PV and the five products of the backward pass) is evaluated as six
// v_mfma_f32_32x32x16_bf16 products of exactly split operands (see gemm_split_kernel.h / split_bf16.h): 6/16 of the matrix
// pipe time of the f32 MFMA at the same accuracy.  Selected by the `arith` argument of ptamd_attention_fwd / _bwd.
//
// Decomposition as in attention.hip but with 8 wavefronts: one workgroup = (protein, head, 256 queries), 32 per
// wavefront (a staged tile is converted once per 256 queries: -10 % against 4-wavefront workgroups), scores
// computed TRANSPOSED (S^T[key][q]) so a softmax row is lane-local and the probability block, still sitting in the
// MFMA accumulator layout, is split in registers and fed straight back as the B operand of O^T += V^T P^T.
// K and V tiles of 32 keys are staged once per workgroup as three bf16 planes each in the swizzled LDS format of
// split_bf16.h, which serves the row-fragment reads (K in QK^T) and the transposed reads (V^T in PV) conflict-free.
#include "attn_dropout.h"
#include "split_bf16.h"

namespace ptattn {
using namespace ptsplit;

constexpr int NTHR = 512;        // 8 wavefronts: a staged tile is converted once for 256 queries (keys) instead of 128
constexpr int QB = NTHR / 2;     // queries (or keys) per workgroup: 32 per wavefront
constexpr int TR = 32;  // rows (keys or queries) of an LDS tile
typedef Tile64<TR> Tile;

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ int crow(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// 32 rows x 64 floats of a [*, ld] matrix, global -> registers -> split planes in LDS (512 threads, 1 float4 each).
// The loads are unconditional (row clamped); rows beyond nrows are zeroed when they are stored.
// DK = 32: a tile row is half as wide (8 float4), the first 256 threads carry the 32 rows; the LDS image keeps the
// 64-wide row format of split_bf16.h with its upper half unused (same swizzle, same conflict-free reads).
template <int DK>
struct Stage32 {
  static constexpr int CPR = DK / 4;  // float4 per tile row
  float4 v[512 / NTHR];
  __device__ __forceinline__ void load(const float *__restrict__ base, int ld, int row0, int nrows, int tid) {
#pragma unroll
    for (int i = 0; i < 512 / NTHR; ++i) {
      const int f = tid + NTHR * i, row = min(row0 + min(f / CPR, TR - 1), nrows - 1);
      v[i] = *reinterpret_cast<const float4 *>(base + (size_t)row * ld + (f % CPR) * 4);
    }
  }
  __device__ __forceinline__ void store(unsigned short *__restrict__ s, int row0, int nrows, int tid) const {
#pragma unroll
    for (int i = 0; i < 512 / NTHR; ++i) {
      const int f = tid + NTHR * i, row = f / CPR;
      if (row < TR) {  // (wavefront-uniform: 64 lanes cover whole rows)
        const bool ok = row0 + row < nrows;
        const float4 x = make_float4(ok ? v[i].x : 0.f, ok ? v[i].y : 0.f, ok ? v[i].z : 0.f, ok ? v[i].w : 0.f);
        Tile::store4(s, row, (f % CPR) * 4, x);
      }
    }
  }
};

// B operands (3 planes x DK / 16 k steps) of one row of a [*, ld] matrix: lane (l31, lh) holds d = 16 s + 8 lh + 0..7
template <int KS>
__device__ __forceinline__ void load_row_split(const float *__restrict__ base, int ld, int row, bool ok, int lh,
                                               bf16x8 (&f)[KS][3]) {
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const float *p = base + (size_t)row * ld + 16 * s + 8 * lh;
    const float4 a = ok ? *reinterpret_cast<const float4 *>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b = ok ? *reinterpret_cast<const float4 *>(p + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    split8(x, f[s]);
  }
}

// =================================================================================================== forward
// LDS: two buffers of {K tile, V tile} (32 keys each), 73.7 KB; one 8-wavefront workgroup per CU (two wavefronts per
// SIMD, 256 VGPRs each).  One barrier per tile: the
// next tile is fetched at the top of the iteration, converted and stored into the other buffer after the scores.
constexpr int BUF = 2 * Tile::ELEMS;  // bf16 elements of one {K, V} buffer
constexpr size_t ATTN_LDS = (size_t)2 * BUF * sizeof(unsigned short);

template <int DK>
__global__ __launch_bounds__(NTHR, 1) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_fwd_split_kernel(const float *__restrict__ qkv, const int64_t *__restrict__ seq,
                                                                int L, int H, float p_drop, uint64_t seed, uint32_t stream_id,
                                                                float *__restrict__ out, float *__restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  __shared__ unsigned int sMask[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QB + wave * 32;
  const int D = H * DK, D3 = 3 * D;
  const float *base = qkv + (size_t)b * L * D3 + h * DK;  // Q block of this head; K at +D, V at +2D
  const int64_t *sq = seq + (size_t)b * L;
  const int q = q0 + l31;
  const bool q_ok = q < L;
  constexpr int KS = DK / 16, NT = DK / 32;  // k steps of Q K^T, 32-wide column tiles of the head dimension
  const float scale = DK == 64 ? 0.125f : 0.17677669529663687f;  // 1 / sqrt(dk)
  const AttnDrop dk_ = make_attn_drop(seed, stream_id, (uint32_t)(b * H + h), p_drop);
  const uint32_t q_part = attn_q_part(dk_, (uint32_t)q);

  bf16x8 qf[KS][3];
  load_row_split(base, D3, min(q, L - 1), q_ok, lh, qf);

  f32x16 o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  Stage32<DK> stK, stV;
  const int ntiles = (L + TR - 1) / TR;
  auto publish_mask = [&](int k0, int buf) __attribute__((always_inline)) {
    if (wave == 0) {
      const int key = k0 + l31;
      const unsigned long long mk = __ballot(lh == 0 && key < L && sq[key < L ? key : 0] != PTAMD_PAD_ID);
      if (lane == 0) sMask[buf] = (unsigned int)mk;
    }
  };
  stK.load(base + D, D3, 0, L, tid);
  stV.load(base + 2 * D, D3, 0, L, tid);
  stK.store(smem, 0, L, tid);
  stV.store(smem + Tile::ELEMS, 0, L, tid);
  publish_mask(0, 0);
  // The loads run TWO tiles ahead of the arithmetic: tile kt + 2 is requested at the top of iteration kt into the second
  // register set while tile kt + 1 (requested an iteration ago, long since arrived) is converted and stored - the wait in
  // front of the conversion no longer exposes the global latency once per tile.  Loads are unconditional (rows clamped):
  // a load behind a branch would make the compiler drain the whole queue at the next use.
  Stage32<DK> nxK, nxV;
  stK.load(base + D, D3, TR, L, tid);
  stV.load(base + 2 * D, D3, TR, L, tid);
  __syncthreads();

  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * TR, cur = kt & 1;
    const bool more = kt + 1 < ntiles;
    const unsigned short *sK = smem + cur * BUF, *sV = sK + Tile::ELEMS;
    nxK.load(base + D, D3, k0 + 2 * TR, L, tid);
    nxV.load(base + 2 * D, D3, k0 + 2 * TR, L, tid);
    const unsigned int mask = sMask[cur];
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int st = 0; st < KS; ++st) {  // S^T[key][q] = K Q^T
      bf16x8 kf[3];
      Tile::frag_rows(sK, 0, st, lane, kf);
      s = mfma6(kf, qf[st], s);
    }
    if (more) {  // convert + store the next tile into the other buffer while the softmax runs
      unsigned short *nK = smem + (cur ^ 1) * BUF;
      stK.store(nK, k0 + TR, L, tid);
      stV.store(nK + Tile::ELEMS, k0 + TR, L, tid);
      publish_mask(k0 + TR, cur ^ 1);
    }
    // masked scores stay UNSCALED here; scale and log2(e) are folded into one fma in front of the exp2:
    // exp(scale s - m) = exp2(c s - c m / scale) with c = scale log2(e) and m kept in scaled units
    constexpr float LOG2E = 1.4426950408889634f;
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool valid = (mask >> crow(r, lh)) & 1u;
      s[r] = valid ? s[r] : -INFINITY;
      mt = fmaxf(mt, s[r]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * scale;
    const float m_new = fmaxf(m_run, mt);
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = fast_exp(m_run - m_safe);
    const float c = scale * LOG2E, mc = -m_safe * LOG2E;
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c, mc));
      ps += s[r];
    }
    l_run = l_run * alpha + ps;
    m_run = m_new;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {  // wave-uniform: after the first tiles the running maximum rarely moves
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {  // scalar multiplies, kept apart: packed f32 VALU stalls the matrix pipe
          float v = o[t][r] * alpha;
          asm volatile("" : "+v"(v));
          o[t][r] = v;
        }
    }
    if (p_drop > 0.f) {
      const uint32_t keep = attn_keep_bits_keys_in_rows(dk_, q_part, k0, lh);
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = (keep >> r) & 1u ? s[r] : 0.f;  // the 1 / (1 - p) is applied to O at the end
    }
    // O^T[d][q] += V^T[d][key] P^T[key][q]: the accumulator rows of s are already in the k order of frag_cols
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float x[8] = {s[8 * m], s[8 * m + 1], s[8 * m + 2], s[8 * m + 3], s[8 * m + 4], s[8 * m + 5], s[8 * m + 6], s[8 * m + 7]};
      bf16x8 pf[3];
      split8(x, pf);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        bf16x8 vf[3];
        Tile::frag_cols(sV, 16 * m, 32 * t, lane, vf);
        o[t] = mfma6(vf, pf, o[t]);
      }
    }
    stK = nxK;
    stV = nxV;
    __syncthreads();  // the other buffer is complete; nobody reads this one any more
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = (p_drop > 0.f ? dk_.ks : 1.f) / l_tot;  // softmax normalisation and the dropout scale in one factor
  if (q_ok) {
    float *op = out + (size_t)(b * L + q) * D + h * DK;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = t * 32 + 8 * g + 4 * lh;
        *reinterpret_cast<float4 *>(op + d) =
            make_float4(o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
      }
    if (lh == 0) lse[((size_t)b * H + h) * L + q] = m_run + logf(l_tot);
  }
}

// =================================================================================================== backward
// dQ: same decomposition as the forward kernel.  Also computes delta[q] = sum_d dO[q,d] O[q,d] and publishes it for the
// dK/dV kernel, which runs after this one on the same stream.
template <int DK>
__global__ __launch_bounds__(NTHR, 1) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dq_split_kernel(const float *__restrict__ qkv, const int64_t *__restrict__ seq,
                                                                   const float *__restrict__ o_fwd, const float *__restrict__ d_o,
                                                                   const float *__restrict__ lse, float *__restrict__ delta,
                                                                   int L, int H, float p_drop, uint64_t seed, uint32_t stream_id,
                                                                   float *__restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  __shared__ unsigned int sMask[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QB + wave * 32;
  const int D = H * DK, D3 = 3 * D;
  const float *base = qkv + (size_t)b * L * D3 + h * DK;
  const int64_t *sq = seq + (size_t)b * L;
  const int q = q0 + l31, qc = min(q, L - 1);
  const bool q_ok = q < L;
  constexpr int KS = DK / 16, NT = DK / 32;
  const float scale = DK == 64 ? 0.125f : 0.17677669529663687f;
  const AttnDrop dk_ = make_attn_drop(seed, stream_id, (uint32_t)(b * H + h), p_drop);
  const uint32_t q_part = attn_q_part(dk_, (uint32_t)q);

  bf16x8 qf[KS][3], gf[KS][3];
  load_row_split(base, D3, qc, q_ok, lh, qf);
  load_row_split(d_o + (size_t)b * L * D + h * DK, D, qc, q_ok, lh, gf);
  const float my_lse = q_ok ? lse[((size_t)b * H + h) * L + q] : 0.f;
  float my_delta = 0.f;
  {  // each lane half holds half of the d of its query's row
    const float *gp = d_o + ((size_t)b * L + qc) * D + h * DK, *op = o_fwd + ((size_t)b * L + qc) * D + h * DK;
#pragma unroll
    for (int st = 0; st < KS; ++st)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float4 g4 = *reinterpret_cast<const float4 *>(gp + 16 * st + 8 * lh + 4 * j);
        const float4 o4 = *reinterpret_cast<const float4 *>(op + 16 * st + 8 * lh + 4 * j);
        my_delta += g4.x * o4.x + g4.y * o4.y + g4.z * o4.z + g4.w * o4.w;
      }
    my_delta += __shfl_xor(my_delta, 32, 64);
    if (!q_ok) my_delta = 0.f;
    if (q_ok && lh == 0) delta[((size_t)b * H + h) * L + q] = my_delta;
  }

  f32x16 dq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;

  Stage32<DK> stK, stV;
  const int ntiles = (L + TR - 1) / TR;
  auto publish_mask = [&](int k0, int buf) __attribute__((always_inline)) {
    if (wave == 0) {
      const int key = k0 + l31;
      const unsigned long long mk = __ballot(lh == 0 && key < L && sq[key < L ? key : 0] != PTAMD_PAD_ID);
      if (lane == 0) sMask[buf] = (unsigned int)mk;
    }
  };
  stK.load(base + D, D3, 0, L, tid);
  stV.load(base + 2 * D, D3, 0, L, tid);
  stK.store(smem, 0, L, tid);
  stV.store(smem + Tile::ELEMS, 0, L, tid);
  publish_mask(0, 0);
  Stage32<DK> nxK, nxV;  // loads two tiles ahead, unconditional (see the forward kernel)
  stK.load(base + D, D3, TR, L, tid);
  stV.load(base + 2 * D, D3, TR, L, tid);
  __syncthreads();

  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * TR, cur = kt & 1;
    const bool more = kt + 1 < ntiles;
    const unsigned short *sK = smem + cur * BUF, *sV = sK + Tile::ELEMS;
    nxK.load(base + D, D3, k0 + 2 * TR, L, tid);
    nxV.load(base + 2 * D, D3, k0 + 2 * TR, L, tid);
    const unsigned int mask = sMask[cur];
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int st = 0; st < KS; ++st) {
      bf16x8 kf[3], vf[3];
      Tile::frag_rows(sK, 0, st, lane, kf);
      Tile::frag_rows(sV, 0, st, lane, vf);
      s = mfma6(kf, qf[st], s);     // S^T[key][q]
      dp = mfma6(vf, gf[st], dp);   // dP^T[key][q] = V dO^T
    }
    if (more) {
      unsigned short *nK = smem + (cur ^ 1) * BUF;
      stK.store(nK, k0 + TR, L, tid);
      stV.store(nK + Tile::ELEMS, k0 + TR, L, tid);
      publish_mask(k0 + TR, cur ^ 1);
    }
    const uint32_t keep = p_drop > 0.f ? attn_keep_bits_keys_in_rows(dk_, q_part, k0, lh) : 0xffffu;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kk = crow(r, lh);
      const bool valid = (mask >> kk) & 1u;
      // (the mask goes into the ARGUMENT, exp2(-inf) = 0: a select around the exp would become a branch per element)
      const float p = fast_exp(valid ? s[r] * scale - my_lse : -INFINITY);
      float g = dp[r];
      if (p_drop > 0.f) g = (keep >> r) & 1u ? g * dk_.ks : 0.f;
      s[r] = p * (g - my_delta) * scale;  // dS^T, already carrying the 1/sqrt(dk) of the scores
    }
    // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float x[8] = {s[8 * m], s[8 * m + 1], s[8 * m + 2], s[8 * m + 3], s[8 * m + 4], s[8 * m + 5], s[8 * m + 6], s[8 * m + 7]};
      bf16x8 df[3];
      split8(x, df);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        bf16x8 kt_[3];
        Tile::frag_cols(sK, 16 * m, 32 * t, lane, kt_);
        dq[t] = mfma6(kt_, df, dq[t]);
      }
    }
    stK = nxK;
    stV = nxV;
    __syncthreads();
  }
  if (q_ok) {
    float *op = dqkv + (size_t)(b * L + q) * D3 + h * DK;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = t * 32 + 8 * g + 4 * lh;
        *reinterpret_cast<float4 *>(op + d) = make_float4(dq[t][4 * g], dq[t][4 * g + 1], dq[t][4 * g + 2], dq[t][4 * g + 3]);
      }
  }
}

// dK, dV: one workgroup = 128 keys of one (protein, head); lane column = key.  The split K and V rows of a lane's key
// stay in registers as B operands; Q and dO tiles of 32 queries stream through LDS and serve both as row fragments
// (S = Q K^T, dP = dO V^T) and as transposed fragments (dK^T += Q^T dS, dV^T += dO^T Pd).
template <int DK>
__global__ __launch_bounds__(NTHR, 1) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dkv_split_kernel(const float *__restrict__ qkv, const int64_t *__restrict__ seq,
                                                                    const float *__restrict__ d_o, const float *__restrict__ lse,
                                                                    const float *__restrict__ delta, int L, int H, float p_drop,
                                                                    uint64_t seed, uint32_t stream_id, float *__restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  __shared__ __attribute__((aligned(16))) float sLse[2][TR], sDel[2][TR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, key0 = blockIdx.x * QB + wave * 32;
  const int D = H * DK, D3 = 3 * D;
  const float *base = qkv + (size_t)b * L * D3 + h * DK;
  const float *gbase = d_o + (size_t)b * L * D + h * DK;
  const float *lse_b = lse + ((size_t)b * H + h) * L, *del_b = delta + ((size_t)b * H + h) * L;
  const int key = key0 + l31;
  const bool k_ok = key < L;
  const bool k_valid = k_ok && seq[(size_t)b * L + key] != PTAMD_PAD_ID;
  constexpr int KS = DK / 16, NT = DK / 32;
  const float scale = DK == 64 ? 0.125f : 0.17677669529663687f;
  const AttnDrop dk_ = make_attn_drop(seed, stream_id, (uint32_t)(b * H + h), p_drop);

  bf16x8 kf[KS][3], vf[KS][3];
  load_row_split(base + D, D3, min(key, L - 1), k_ok, lh, kf);
  load_row_split(base + 2 * D, D3, min(key, L - 1), k_ok, lh, vf);

  f32x16 dk[NT], dv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[t][r] = dv[t][r] = 0.f;

  Stage32<DK> stQ, stG;
  const int ntiles = (L + TR - 1) / TR;
  float r_lse = 0.f, r_del = 0.f;
  stQ.load(base, D3, 0, L, tid);
  stG.load(gbase, D, 0, L, tid);
  stQ.store(smem, 0, L, tid);
  stG.store(smem + Tile::ELEMS, 0, L, tid);
  if (tid < TR) {
    sLse[0][tid] = tid < L ? lse_b[tid] : 0.f;
    sDel[0][tid] = tid < L ? del_b[tid] : 0.f;
  }
  Stage32<DK> nxQ, nxG;  // loads two tiles ahead, unconditional (see the forward kernel)
  stQ.load(base, D3, TR, L, tid);
  stG.load(gbase, D, TR, L, tid);
  __syncthreads();

  for (int qt = 0; qt < ntiles; ++qt) {
    const int qq0 = qt * TR, cur = qt & 1;
    const bool more = qt + 1 < ntiles;
    const unsigned short *sQ = smem + cur * BUF, *sG = sQ + Tile::ELEMS;
    nxQ.load(base, D3, qq0 + 2 * TR, L, tid);
    nxG.load(gbase, D, qq0 + 2 * TR, L, tid);
    if (more) {
      if (tid < TR) {
        const int qn = qq0 + TR + tid;
        r_lse = qn < L ? lse_b[qn] : 0.f;
        r_del = qn < L ? del_b[qn] : 0.f;
      }
    }
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int st = 0; st < KS; ++st) {
      bf16x8 qa[3], ga[3];
      Tile::frag_rows(sQ, 0, st, lane, qa);
      Tile::frag_rows(sG, 0, st, lane, ga);
      s = mfma6(qa, kf[st], s);     // S[q][key]
      dp = mfma6(ga, vf[st], dp);   // dP[q][key] = dO V^T
    }
    if (more) {
      unsigned short *nQ = smem + (cur ^ 1) * BUF;
      stQ.store(nQ, qq0 + TR, L, tid);
      stG.store(nQ + Tile::ELEMS, qq0 + TR, L, tid);
      if (tid < TR) {
        sLse[cur ^ 1][tid] = r_lse;
        sDel[cur ^ 1][tid] = r_del;
      }
    }
    f32x16 pd;  // dropped probabilities (operand of dV)
    const uint32_t keepbits = p_drop > 0.f ? attn_keep_bits_queries_in_rows(dk_, (uint32_t)key, qq0, lh) : 0xffffu;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = crow(r, lh), qg = qq0 + qi;
      const bool ok = k_valid && qg < L;
      const float my_l = sLse[cur][qi], my_d = sDel[cur][qi];  // (LDS broadcast reads: registers are the scarce resource here)
      // (the mask goes into the ARGUMENT, exp2(-inf) = 0: a select around the exp would become a branch per element)
      const float p = fast_exp(ok ? s[r] * scale - my_l : -INFINITY);
      float g = dp[r], pk = p;
      if (p_drop > 0.f) {
        const bool keep = (keepbits >> r) & 1u;
        g = keep ? g * dk_.ks : 0.f;
        pk = keep ? p * dk_.ks : 0.f;
      }
      pd[r] = pk;
      s[r] = p * (g - my_d) * scale;  // dS[q][key]
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float xs[8] = {s[8 * m], s[8 * m + 1], s[8 * m + 2], s[8 * m + 3], s[8 * m + 4], s[8 * m + 5], s[8 * m + 6], s[8 * m + 7]};
      const float xp[8] = {pd[8 * m], pd[8 * m + 1], pd[8 * m + 2], pd[8 * m + 3], pd[8 * m + 4], pd[8 * m + 5], pd[8 * m + 6], pd[8 * m + 7]};
      bf16x8 dsf[3], pdf[3];
      split8(xs, dsf);
      split8(xp, pdf);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        bf16x8 qt_[3], gt_[3];
        Tile::frag_cols(sQ, 16 * m, 32 * t, lane, qt_);
        dk[t] = mfma6(qt_, dsf, dk[t]);   // dK^T[d][key] += Q^T dS
        Tile::frag_cols(sG, 16 * m, 32 * t, lane, gt_);
        dv[t] = mfma6(gt_, pdf, dv[t]);   // dV^T[d][key] += dO^T Pd
      }
    }
    stQ = nxQ;
    stG = nxG;
    __syncthreads();
  }
  if (k_ok) {
    float *okp = dqkv + (size_t)(b * L + key) * D3 + D + h * DK, *ovp = okp + D;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = t * 32 + 8 * g + 4 * lh;
        *reinterpret_cast<float4 *>(okp + d) = make_float4(dk[t][4 * g], dk[t][4 * g + 1], dk[t][4 * g + 2], dk[t][4 * g + 3]);
        *reinterpret_cast<float4 *>(ovp + d) = make_float4(dv[t][4 * g], dv[t][4 * g + 1], dv[t][4 * g + 2], dv[t][4 * g + 3]);
      }
  }
}

template <typename Kern>
int set_lds(Kern kern) {
  PT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)ATTN_LDS));
  return PTAMD_OK;
}

}  // namespace ptattn

int pt_attention_fwd_split(const float *qkv, const int64_t *seq, int B, int L, int H, int dk, float p, uint64_t seed,
                           uint32_t sid, float *out, float *lse, hipStream_t st) {
  using namespace ptattn;
  const dim3 grid((L + QB - 1) / QB, H, B);
  if (dk == 64) {
    if (int rc = set_lds(attn_fwd_split_kernel<64>)) return rc;  // idempotent, host-only: no state kept between calls
    hipLaunchKernelGGL(attn_fwd_split_kernel<64>, grid, dim3(NTHR), ATTN_LDS, st, qkv, seq, L, H, p, seed, sid, out, lse);
  } else {
    if (int rc = set_lds(attn_fwd_split_kernel<32>)) return rc;
    hipLaunchKernelGGL(attn_fwd_split_kernel<32>, grid, dim3(NTHR), ATTN_LDS, st, qkv, seq, L, H, p, seed, sid, out, lse);
  }
  return pt_check_launch();
}

int pt_attention_bwd_split(const float *qkv, const int64_t *seq, const float *o_fwd, const float *d_o, const float *lse,
                           float *delta, int B, int L, int H, int dk, float p, uint64_t seed, uint32_t sid, float *dqkv,
                           hipStream_t st) {
  using namespace ptattn;
  const dim3 grid((L + QB - 1) / QB, H, B);
  if (dk == 64) {
    if (int rc = set_lds(attn_bwd_dq_split_kernel<64>)) return rc;
    if (int rc = set_lds(attn_bwd_dkv_split_kernel<64>)) return rc;
    hipLaunchKernelGGL(attn_bwd_dq_split_kernel<64>, grid, dim3(NTHR), ATTN_LDS, st, qkv, seq, o_fwd, d_o, lse, delta, L, H, p,
                       seed, sid, dqkv);
    if (int rc = pt_check_launch()) return rc;
    hipLaunchKernelGGL(attn_bwd_dkv_split_kernel<64>, grid, dim3(NTHR), ATTN_LDS, st, qkv, seq, d_o, lse, delta, L, H, p, seed,
                       sid, dqkv);
  } else {
    if (int rc = set_lds(attn_bwd_dq_split_kernel<32>)) return rc;
    if (int rc = set_lds(attn_bwd_dkv_split_kernel<32>)) return rc;
    hipLaunchKernelGGL(attn_bwd_dq_split_kernel<32>, grid, dim3(NTHR), ATTN_LDS, st, qkv, seq, o_fwd, d_o, lse, delta, L, H, p,
                       seed, sid, dqkv);
    if (int rc = pt_check_launch()) return rc;
    hipLaunchKernelGGL(attn_bwd_dkv_split_kernel<32>, grid, dim3(NTHR), ATTN_LDS, st, qkv, seq, d_o, lse, delta, L, H, p, seed,
                       sid, dqkv);
  }
  return pt_check_launch();
}
