// Fused masked multi-head self-attention, forward and backward, on the f32 matrix cores of gfx950.
//
// Replaces ScaledDotProductAttention + the head split/merge of MultiHeadedAttention
//   /root/reference/protein_transformer/models/transformer/Attention.py:14-22,55-68
// (scores = QK^T / sqrt(dk); masked_fill(key is padding, -inf); softmax; dropout(p) on the probabilities; P V)
// without ever materialising the [B,H,L,L] score tensor (268 MB at B=32, L=512) and its autograd copies.
//
// Design (wave64, v_mfma_f32_32x32x2_f32, exact f32):
//   * forward / dQ kernels: one workgroup = (protein, head, 128 queries), 4 wavefronts x 32 queries.  K/V tiles
//     of 64 keys are staged once per workgroup in LDS (register double buffering, one barrier per tile).
//     Scores are computed TRANSPOSED, S^T[key][q] = mfma(K, Q): the MFMA C layout then puts one query per
//     lane column, so the online-softmax row statistics are lane-local (16 registers + one cross-half
//     shuffle) and P^T is already the B operand of the next product, O^T[d][q] += mfma(V^T, P^T).
//   * dK/dV kernel: one workgroup = (protein, head, 128 keys); K, V fragments live in registers, Q / dO
//     tiles of 32 queries stream through LDS; scores are computed untransposed so that one KEY sits in
//     each lane column and dK^T, dV^T accumulate in registers.  No atomics anywhere: results are
//     deterministic.
//   * dropout masks come from a counter hash of (seed, protein*head, query, key) and are regenerated in the
//     backward kernels instead of being stored.
#include "attn_dropout.h"
#include "kv_format.h"

// split-bf16 variants for dk = 64 and dk = 32 (attention_split.hip), selected by the `arith` argument of the entry points
int pt_attention_fwd_split(const float *qkv, const int64_t *seq, int B, int L, int H, int dk, float p, uint64_t seed,
                           uint32_t sid, float *out, float *lse, hipStream_t st);
int pt_attention_bwd_split(const float *qkv, const int64_t *seq, const float *o_fwd, const float *d_o, const float *lse,
                           float *delta, int B, int L, int H, int dk, float p, uint64_t seed, uint32_t sid, float *dqkv,
                           hipStream_t st);
// two-term f16 variants (attention_f16x2.hip): arith = PTAMD_GEMM_AUTO / PTAMD_GEMM_F16X2
int pt_attention_fwd_f16x2(const float *qkv, const int64_t *seq, int B, int L, int H, int dk, float p, uint64_t seed,
                           uint32_t sid, float *out, float *lse, uint32_t *keep_bits, const void *kv_planes, const float *kv_inv,
                           hipStream_t st);
int pt_attention_bwd_f16x2(const float *qkv, const int64_t *seq, const float *o_fwd, const float *d_o, const float *lse,
                           float *delta, int B, int L, int H, int dk, float p, uint64_t seed, uint32_t sid, float *dqkv,
                           uint32_t *row_scale, uint32_t *row_min, const uint32_t *keep_bits, const void *kv_planes,
                           const float *kv_inv, float *slabs, size_t slab_floats, hipStream_t st);
size_t pt_attention_bwd_f16x2_slab_floats(int B, int L, int H, int dk);
bool pt_attention_bwd_f16x2_reads_keep_bits(int B, int L, int H, int dk);
bool pt_attention_f16x2_reads_kv_planes(int B, int L, int H, int dk);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int QB = 128;  // queries (or keys) per workgroup
constexpr int KT = 64;   // keys per LDS tile in the forward / dQ kernels
constexpr int QT = 32;   // queries per LDS tile in the dK/dV kernel

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
// row index inside a 32x32 MFMA C tile held by (register r, lane half lh)
__device__ __forceinline__ int crow(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// ---- cooperative tile staging: rows x DK floats from a [*, ld] matrix into an LDS image with stride DK+1
template <int DK, int ROWS>
struct Stage {
  static constexpr int F4 = ROWS * DK / 4;            // float4 per tile
  static constexpr int PER = (F4 + 255) / 256;        // per thread
  float4 v[PER];
  __device__ __forceinline__ void load(const float *__restrict__ base, int ld, int row0, int nrows, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int f = tid + 256 * i;
      const int row = f / (DK / 4), c = (f % (DK / 4)) * 4;
      v[i] = (f < F4 && row0 + row < nrows) ? *reinterpret_cast<const float4 *>(base + (size_t)(row0 + row) * ld + c)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void store(float *__restrict__ s, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int f = tid + 256 * i;
      if (f < F4) {
        const int row = f / (DK / 4), c = (f % (DK / 4)) * 4;
        float *d = s + row * (DK + 1) + c;
        d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
      }
    }
  }
};

// a lane's MFMA B fragment of a [32 rows x DK] global matrix block: frag[s] = M[row0 + l31][2s + lh]
template <int DK>
__device__ __forceinline__ void load_row_frag(const float *__restrict__ base, int ld, int row, bool ok, int lh,
                                              float (&frag)[DK / 2]) {
#pragma unroll
  for (int j = 0; j < DK / 4; ++j) {
    const float4 v = ok ? *reinterpret_cast<const float4 *>(base + (size_t)row * ld + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    frag[2 * j] = lh ? v.y : v.x;
    frag[2 * j + 1] = lh ? v.w : v.z;
  }
}

// =================================================================================================== forward
template <int DK>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const float *__restrict__ qkv, const int64_t *__restrict__ seq,
                                                       int L, int H, float p_drop, uint64_t seed, uint32_t stream_id,
                                                       float *__restrict__ out, float *__restrict__ lse) {
  constexpr int LDK = DK + 1, NS = DK / 2, NDT = (DK + 31) / 32;
  __shared__ float sK[2][KT * LDK];
  __shared__ float sV[2][KT * LDK];
  __shared__ unsigned long long sMask[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QB + wave * 32;
  const int D = H * DK, D3 = 3 * D;
  const float *base = qkv + (size_t)b * L * D3 + h * DK;  // Q block of this head; K at +D, V at +2D
  const int64_t *sq = seq + (size_t)b * L;
  const int q = q0 + l31;
  const bool q_ok = q < L;
  const float scale = 1.f / sqrtf((float)DK);
  const AttnDrop dk_ = make_attn_drop(seed, stream_id, (uint32_t)(b * H + h), p_drop);
  const uint32_t q_part = attn_q_part(dk_, (uint32_t)q);

  float qf[NS];
  load_row_frag<DK>(base, D3, q, q_ok, lh, qf);

  f32x16 o[NDT];
#pragma unroll
  for (int t = 0; t < NDT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  Stage<DK, KT> stK, stV;
  const int ntiles = (L + KT - 1) / KT;
  stK.load(base + D, D3, 0, L, tid);
  stV.load(base + 2 * D, D3, 0, L, tid);
  stK.store(sK[0], tid);
  stV.store(sV[0], tid);
  if (wave == 0) {
    const unsigned long long mk = __ballot(lane < L && sq[lane < L ? lane : 0] != PTAMD_PAD_ID);
    if (lane == 0) sMask[0] = mk;
  }
  __syncthreads();

  for (int kt = 0; kt < ntiles; ++kt) {
    const int cur = kt & 1, k0 = kt * KT;
    const bool more = kt + 1 < ntiles;
    if (more) {
      stK.load(base + D, D3, k0 + KT, L, tid);
      stV.load(base + 2 * D, D3, k0 + KT, L, tid);
    }
    const unsigned long long mask = sMask[cur];
    const float *tK = sK[cur], *tV = sV[cur];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      if (k0 + sub * 32 >= L) break;
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
      const float *kp = tK + (sub * 32 + l31) * LDK + lh;
#pragma unroll
      for (int st = 0; st < NS; ++st) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * st], qf[st], s, 0, 0, 0);
      // masked, scaled scores; lane column = query, registers = 16 of the 32 keys
      float mt = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kk = sub * 32 + crow(r, lh);
        const bool valid = (mask >> kk) & 1ull;
        s[r] = valid ? s[r] * scale : -INFINITY;
        mt = fmaxf(mt, s[r]);
      }
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float m_new = fmaxf(m_run, mt);
      const float m_safe = m_new == -INFINITY ? 0.f : m_new;
      const float alpha = fast_exp(m_run - m_safe);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = fast_exp(s[r] - m_safe);
        ps += s[r];
      }
      l_run = l_run * alpha + ps;
      m_run = m_new;
#pragma unroll
      for (int t = 0; t < NDT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      if (p_drop > 0.f) {
        const uint32_t keep = attn_keep_bits_keys_in_rows(dk_, q_part, k0 + sub * 32, lh);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = (keep >> r) & 1u ? s[r] : 0.f;  // the 1 / (1 - p) is applied to O at the end
      }
      // O^T[d][q] += V^T[d][key] P^T[key][q]
#pragma unroll
      for (int t = 0; t < NDT; ++t) {
        const int d = t * 32 + l31;
        const float *vp = tV + (sub * 32 + 4 * lh) * LDK + (d < DK ? d : 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float a = d < DK ? vp[((r & 3) + 8 * (r >> 2)) * LDK] : 0.f;
          o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[r], o[t], 0, 0, 0);
        }
      }
    }
    if (more) {
      stK.store(sK[cur ^ 1], tid);
      stV.store(sV[cur ^ 1], tid);
      if (wave == 0) {
        const int key = k0 + KT + lane;
        const unsigned long long mk = __ballot(key < L && sq[key < L ? key : 0] != PTAMD_PAD_ID);
        if (lane == 0) sMask[cur ^ 1] = mk;
      }
    }
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = (p_drop > 0.f ? dk_.ks : 1.f) / l_tot;  // softmax normalisation and the dropout scale in one factor
  if (q_ok) {
    float *op = out + (size_t)(b * L + q) * D + h * DK;
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = t * 32 + 8 * g + 4 * lh;
        if (d < DK)
          *reinterpret_cast<float4 *>(op + d) =
              make_float4(o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
      }
    if (lh == 0) lse[((size_t)b * H + h) * L + q] = m_run + logf(l_tot);
  }
}

// =================================================================================================== backward
// dQ: same decomposition as the forward kernel
template <int DK>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const float *__restrict__ qkv, const int64_t *__restrict__ seq,
                                                          const float *__restrict__ o_fwd, const float *__restrict__ d_o,
                                                          const float *__restrict__ lse, float *__restrict__ delta, int L,
                                                          int H, float p_drop, uint64_t seed, uint32_t stream_id,
                                                          float *__restrict__ dqkv) {
  constexpr int LDK = DK + 1, NS = DK / 2, NDT = (DK + 31) / 32;
  __shared__ float sK[2][KT * LDK];
  __shared__ float sV[2][KT * LDK];
  __shared__ unsigned long long sMask[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QB + wave * 32;
  const int D = H * DK, D3 = 3 * D;
  const float *base = qkv + (size_t)b * L * D3 + h * DK;
  const int64_t *sq = seq + (size_t)b * L;
  const int q = q0 + l31;
  const bool q_ok = q < L;
  const float scale = 1.f / sqrtf((float)DK);
  const AttnDrop dk_ = make_attn_drop(seed, stream_id, (uint32_t)(b * H + h), p_drop);
  const uint32_t q_part = attn_q_part(dk_, (uint32_t)q);

  float qf[NS], gf[NS];
  load_row_frag<DK>(base, D3, q, q_ok, lh, qf);
  load_row_frag<DK>(d_o + (size_t)b * L * D + h * DK, D, q, q_ok, lh, gf);
  const float my_lse = q_ok ? lse[((size_t)b * H + h) * L + q] : 0.f;
  // delta[q] = sum_d dO[q,d] O[q,d]: each lane half holds every other d of its query's row; published for the
  // dK/dV kernel, which runs after this one on the same stream
  float my_delta = 0.f;
  {
    float of[NS];
    load_row_frag<DK>(o_fwd + (size_t)b * L * D + h * DK, D, q, q_ok, lh, of);
#pragma unroll
    for (int st = 0; st < NS; ++st) my_delta += gf[st] * of[st];
    my_delta += __shfl_xor(my_delta, 32, 64);
    if (q_ok && lh == 0) delta[((size_t)b * H + h) * L + q] = my_delta;
  }

  f32x16 dq[NDT];
#pragma unroll
  for (int t = 0; t < NDT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;

  Stage<DK, KT> stK, stV;
  const int ntiles = (L + KT - 1) / KT;
  stK.load(base + D, D3, 0, L, tid);
  stV.load(base + 2 * D, D3, 0, L, tid);
  stK.store(sK[0], tid);
  stV.store(sV[0], tid);
  if (wave == 0) {
    const unsigned long long mk = __ballot(lane < L && sq[lane < L ? lane : 0] != PTAMD_PAD_ID);
    if (lane == 0) sMask[0] = mk;
  }
  __syncthreads();

  for (int kt = 0; kt < ntiles; ++kt) {
    const int cur = kt & 1, k0 = kt * KT;
    const bool more = kt + 1 < ntiles;
    if (more) {
      stK.load(base + D, D3, k0 + KT, L, tid);
      stV.load(base + 2 * D, D3, k0 + KT, L, tid);
    }
    const unsigned long long mask = sMask[cur];
    const float *tK = sK[cur], *tV = sV[cur];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      if (k0 + sub * 32 >= L) break;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
      const float *kp = tK + (sub * 32 + l31) * LDK + lh;
      const float *vp = tV + (sub * 32 + l31) * LDK + lh;
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * st], qf[st], s, 0, 0, 0);    // S^T[key][q]
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[2 * st], gf[st], dp, 0, 0, 0);  // dP^T[key][q] = V dO^T
      }
      const uint32_t keep = p_drop > 0.f ? attn_keep_bits_keys_in_rows(dk_, q_part, k0 + sub * 32, lh) : 0xffffu;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kk = sub * 32 + crow(r, lh);
        const bool valid = (mask >> kk) & 1ull;
        const float p = valid ? fast_exp(s[r] * scale - my_lse) : 0.f;
        float g = dp[r];
        if (p_drop > 0.f) g = (keep >> r) & 1u ? g * dk_.ks : 0.f;
        s[r] = p * (g - my_delta) * scale;  // dS^T, already carrying the 1/sqrt(dk) of the scores
      }
      // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
      for (int t = 0; t < NDT; ++t) {
        const int d = t * 32 + l31;
        const float *kq = tK + (sub * 32 + 4 * lh) * LDK + (d < DK ? d : 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float a = d < DK ? kq[((r & 3) + 8 * (r >> 2)) * LDK] : 0.f;
          dq[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[r], dq[t], 0, 0, 0);
        }
      }
    }
    if (more) {
      stK.store(sK[cur ^ 1], tid);
      stV.store(sV[cur ^ 1], tid);
      if (wave == 0) {
        const int key = k0 + KT + lane;
        const unsigned long long mk = __ballot(key < L && sq[key < L ? key : 0] != PTAMD_PAD_ID);
        if (lane == 0) sMask[cur ^ 1] = mk;
      }
    }
    __syncthreads();
  }
  if (q_ok) {
    float *op = dqkv + (size_t)(b * L + q) * D3 + h * DK;
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = t * 32 + 8 * g + 4 * lh;
        if (d < DK)
          *reinterpret_cast<float4 *>(op + d) = make_float4(dq[t][4 * g], dq[t][4 * g + 1], dq[t][4 * g + 2], dq[t][4 * g + 3]);
      }
  }
}

// dK, dV: one workgroup = 128 keys of one (protein, head); lane column = key
template <int DK>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const float *__restrict__ qkv, const int64_t *__restrict__ seq,
                                                           const float *__restrict__ d_o, const float *__restrict__ lse,
                                                           const float *__restrict__ delta, int L, int H, float p_drop,
                                                           uint64_t seed, uint32_t stream_id, float *__restrict__ dqkv) {
  constexpr int LDK = DK + 1, NS = DK / 2, NDT = (DK + 31) / 32;
  __shared__ float sQ[2][QT * LDK];
  __shared__ float sG[2][QT * LDK];
  __shared__ float sLse[2][QT], sDel[2][QT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, key0 = blockIdx.x * QB + wave * 32;
  const int D = H * DK, D3 = 3 * D;
  const float *base = qkv + (size_t)b * L * D3 + h * DK;
  const float *gbase = d_o + (size_t)b * L * D + h * DK;
  const float *lse_b = lse + ((size_t)b * H + h) * L, *del_b = delta + ((size_t)b * H + h) * L;
  const int key = key0 + l31;
  const bool k_ok = key < L;
  const bool k_valid = k_ok && seq[(size_t)b * L + key] != PTAMD_PAD_ID;
  const float scale = 1.f / sqrtf((float)DK);
  const AttnDrop dk_ = make_attn_drop(seed, stream_id, (uint32_t)(b * H + h), p_drop);

  float kf[NS], vf[NS];
  load_row_frag<DK>(base + D, D3, key, k_ok, lh, kf);
  load_row_frag<DK>(base + 2 * D, D3, key, k_ok, lh, vf);

  f32x16 dk[NDT], dv[NDT];
#pragma unroll
  for (int t = 0; t < NDT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[t][r] = dv[t][r] = 0.f;

  Stage<DK, QT> stQ, stG;
  const int ntiles = (L + QT - 1) / QT;
  float r_lse = 0.f, r_del = 0.f;
  stQ.load(base, D3, 0, L, tid);
  stG.load(gbase, D, 0, L, tid);
  stQ.store(sQ[0], tid);
  stG.store(sG[0], tid);
  if (tid < QT) {
    sLse[0][tid] = tid < L ? lse_b[tid] : 0.f;
    sDel[0][tid] = tid < L ? del_b[tid] : 0.f;
  }
  __syncthreads();

  for (int qt = 0; qt < ntiles; ++qt) {
    const int cur = qt & 1, qq0 = qt * QT;
    const bool more = qt + 1 < ntiles;
    if (more) {
      stQ.load(base, D3, qq0 + QT, L, tid);
      stG.load(gbase, D, qq0 + QT, L, tid);
      if (tid < QT) {
        const int qn = qq0 + QT + tid;
        r_lse = qn < L ? lse_b[qn] : 0.f;
        r_del = qn < L ? del_b[qn] : 0.f;
      }
    }
    const float *tQ = sQ[cur], *tG = sG[cur];
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
    const float *qp = tQ + l31 * LDK + lh, *gp = tG + l31 * LDK + lh;
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(qp[2 * st], kf[st], s, 0, 0, 0);    // S[q][key]
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(gp[2 * st], vf[st], dp, 0, 0, 0);  // dP[q][key] = dO V^T
    }
    f32x16 pd;  // dropped probabilities (operand of dV)
    const uint32_t keepbits = p_drop > 0.f ? attn_keep_bits_queries_in_rows(dk_, (uint32_t)key, qq0, lh) : 0xffffu;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = crow(r, lh), qg = qq0 + qi;
      const bool ok = k_valid && qg < L;
      const float p = ok ? fast_exp(s[r] * scale - sLse[cur][qi]) : 0.f;
      float g = dp[r], pk = p;
      if (p_drop > 0.f) {
        const bool keep = (keepbits >> r) & 1u;
        g = keep ? g * dk_.ks : 0.f;
        pk = keep ? p * dk_.ks : 0.f;
      }
      pd[r] = pk;
      s[r] = p * (g - sDel[cur][qi]) * scale;  // dS[q][key]
    }
#pragma unroll
    for (int t = 0; t < NDT; ++t) {
      const int d = t * 32 + l31;
      const float *qd = tQ + (4 * lh) * LDK + (d < DK ? d : 0);
      const float *gd = tG + (4 * lh) * LDK + (d < DK ? d : 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ro = ((r & 3) + 8 * (r >> 2)) * LDK;
        const float aq = d < DK ? qd[ro] : 0.f, ag = d < DK ? gd[ro] : 0.f;
        dk[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq, s[r], dk[t], 0, 0, 0);   // dK^T[d][key] += Q^T dS
        dv[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ag, pd[r], dv[t], 0, 0, 0);  // dV^T[d][key] += dO^T Pd
      }
    }
    if (more) {
      stQ.store(sQ[cur ^ 1], tid);
      stG.store(sG[cur ^ 1], tid);
      if (tid < QT) {
        sLse[cur ^ 1][tid] = r_lse;
        sDel[cur ^ 1][tid] = r_del;
      }
    }
    __syncthreads();
  }
  if (k_ok) {
    float *okp = dqkv + (size_t)(b * L + key) * D3 + D + h * DK, *ovp = okp + D;
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = t * 32 + 8 * g + 4 * lh;
        if (d < DK) {
          *reinterpret_cast<float4 *>(okp + d) = make_float4(dk[t][4 * g], dk[t][4 * g + 1], dk[t][4 * g + 2], dk[t][4 * g + 3]);
          *reinterpret_cast<float4 *>(ovp + d) = make_float4(dv[t][4 * g], dv[t][4 * g + 1], dv[t][4 * g + 2], dv[t][4 * g + 3]);
        }
      }
  }
}

template <int DK>
int launch_fwd(const float *qkv, const int64_t *seq, int B, int L, int H, float p, uint64_t seed, uint32_t sid,
               float *out, float *lse, hipStream_t st) {
  hipLaunchKernelGGL(attn_fwd_kernel<DK>, dim3((L + QB - 1) / QB, H, B), dim3(256), 0, st, qkv, seq, L, H, p, seed, sid,
                     out, lse);
  return pt_check_launch();
}
template <int DK>
int launch_bwd(const float *qkv, const int64_t *seq, const float *o_fwd, const float *d_o, const float *lse, float *delta,
               int B, int L, int H, float p, uint64_t seed, uint32_t sid, float *dqkv, hipStream_t st) {
  const dim3 grid((L + QB - 1) / QB, H, B);
  hipLaunchKernelGGL(attn_bwd_dq_kernel<DK>, grid, dim3(256), 0, st, qkv, seq, o_fwd, d_o, lse, delta, L, H, p, seed, sid,
                     dqkv);
  int rc = pt_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(attn_bwd_dkv_kernel<DK>, grid, dim3(256), 0, st, qkv, seq, d_o, lse, delta, L, H, p, seed, sid, dqkv);
  return pt_check_launch();
}

}  // namespace

extern "C" {

namespace {
size_t delta_floats(int B, int L, int H) { return ((size_t)B * H * L + 3) & ~(size_t)3; }   // (what follows stays 16-byte aligned)
}
size_t ptamd_attention_workspace_bytes(int B, int L, int H, int dk) {
  if (B <= 0 || L <= 0 || H <= 0) return 0;
  // delta [B, H, L]; behind it the slabs of the split one-sweep backward kernel where this shape takes it (head size 64, few
  // (protein, head) pairs: csrc/attention_f16x2.hip fused_split)
  return (delta_floats(B, L, H) + pt_attention_bwd_f16x2_slab_floats(B, L, H, dk)) * sizeof(float);
}

size_t ptamd_attention_keep_bits_bytes(int B, int L, int H) {
  if (B <= 0 || L <= 0 || H <= 0) return 0;
  return attn_keep_words(B, L, H) * sizeof(uint32_t);
}

int ptamd_attention_bwd_reads_keep_bits(int B, int L, int H, int dk, int arith) {
  if (B <= 0 || L <= 0 || H <= 0 || !(arith == PTAMD_GEMM_AUTO || arith == PTAMD_GEMM_F16X2)) return 0;
  return pt_attention_bwd_f16x2_reads_keep_bits(B, L, H, dk) ? 1 : 0;
}

size_t ptamd_attention_kv_bytes(int T, int H) { return (T <= 0 || H <= 0) ? 0 : ptkv::planes_bytes(T, H); }
size_t ptamd_attention_kv_inv_floats(int T, int H) { return (T <= 0 || H <= 0) ? 0 : ptkv::inv_floats(T, H); }
int ptamd_attention_reads_kv_planes(int B, int L, int H, int dk, int arith) {
  if (B <= 0 || L <= 0 || H <= 0 || !(arith == PTAMD_GEMM_AUTO || arith == PTAMD_GEMM_F16X2)) return 0;
  return pt_attention_f16x2_reads_kv_planes(B, L, H, dk) ? 1 : 0;
}

int ptamd_attention_fwd(const float *qkv, const int64_t *seq, int B, int L, int H, int dk, float dropout_p,
                        uint64_t seed, uint32_t stream_id, int arith, float *out, float *lse, uint32_t *keep_bits,
                        const void *kv_planes, const float *kv_inv, void *stream) {
  if (B <= 0 || L <= 0 || H <= 0 || arith < PTAMD_GEMM_F32 || arith > PTAMD_GEMM_AUTO) return PTAMD_ERR_BAD_SHAPE;
  if (dropout_p < 0.f || dropout_p >= 1.f) return PTAMD_ERR_BAD_SHAPE;
  if (!pt_aligned16(qkv) || !pt_aligned16(out)) return PTAMD_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const bool f16x2 = (dk == 64 || dk == 32) && (arith == PTAMD_GEMM_AUTO || arith == PTAMD_GEMM_F16X2);
  if (keep_bits && !f16x2) return PTAMD_ERR_BAD_SHAPE;  // the decisions are a by-product of the f16x2 kernels only
  if (kv_planes && !(f16x2 && kv_inv && pt_aligned16(kv_planes))) return PTAMD_ERR_BAD_SHAPE;   // ... the planes their input only
  if (f16x2) return pt_attention_fwd_f16x2(qkv, seq, B, L, H, dk, dropout_p, seed, stream_id, out, lse, keep_bits, kv_planes, kv_inv, st);
  if ((dk == 64 || dk == 32) && arith != PTAMD_GEMM_F32)
    return pt_attention_fwd_split(qkv, seq, B, L, H, dk, dropout_p, seed, stream_id, out, lse, st);
  switch (dk) {
    case 8: return launch_fwd<8>(qkv, seq, B, L, H, dropout_p, seed, stream_id, out, lse, st);
    case 16: return launch_fwd<16>(qkv, seq, B, L, H, dropout_p, seed, stream_id, out, lse, st);
    case 32: return launch_fwd<32>(qkv, seq, B, L, H, dropout_p, seed, stream_id, out, lse, st);
    case 64: return launch_fwd<64>(qkv, seq, B, L, H, dropout_p, seed, stream_id, out, lse, st);
    default: return PTAMD_ERR_BAD_SHAPE;
  }
}

int ptamd_attention_bwd(const float *qkv, const int64_t *seq, const float *out, const float *dout, const float *lse,
                        int B, int L, int H, int dk, float dropout_p, uint64_t seed, uint32_t stream_id, int arith,
                        float *dqkv, uint32_t *row_scale, uint32_t *row_scale_min, const uint32_t *keep_bits,
                        const void *kv_planes, const float *kv_inv, void *workspace, size_t workspace_bytes, void *stream) {
  if (B <= 0 || L <= 0 || H <= 0 || arith < PTAMD_GEMM_F32 || arith > PTAMD_GEMM_AUTO) return PTAMD_ERR_BAD_SHAPE;
  if (dropout_p < 0.f || dropout_p >= 1.f) return PTAMD_ERR_BAD_SHAPE;
  if (!workspace || workspace_bytes < ptamd_attention_workspace_bytes(B, L, H, dk)) return PTAMD_ERR_WORKSPACE;
  if (!pt_aligned16(qkv) || !pt_aligned16(out) || !pt_aligned16(dout) || !pt_aligned16(dqkv)) return PTAMD_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  float *delta = static_cast<float *>(workspace);
  if ((dk == 64 || dk == 32) && (arith == PTAMD_GEMM_AUTO || arith == PTAMD_GEMM_F16X2))
    return pt_attention_bwd_f16x2(qkv, seq, out, dout, lse, delta, B, L, H, dk, dropout_p, seed, stream_id, dqkv, row_scale,
                                  row_scale ? row_scale_min : nullptr, keep_bits, kv_planes, kv_inv, delta + delta_floats(B, L, H),
                                  workspace_bytes / sizeof(float) - delta_floats(B, L, H), st);
  if (row_scale || row_scale_min || keep_bits || kv_planes) return PTAMD_ERR_BAD_SHAPE;  // by-products / inputs of the f16x2 kernels only
  if ((dk == 64 || dk == 32) && arith != PTAMD_GEMM_F32)
    return pt_attention_bwd_split(qkv, seq, out, dout, lse, delta, B, L, H, dk, dropout_p, seed, stream_id, dqkv, st);
  switch (dk) {
    case 8: return launch_bwd<8>(qkv, seq, out, dout, lse, delta, B, L, H, dropout_p, seed, stream_id, dqkv, st);
    case 16: return launch_bwd<16>(qkv, seq, out, dout, lse, delta, B, L, H, dropout_p, seed, stream_id, dqkv, st);
    case 32: return launch_bwd<32>(qkv, seq, out, dout, lse, delta, B, L, H, dropout_p, seed, stream_id, dqkv, st);
    case 64: return launch_bwd<64>(qkv, seq, out, dout, lse, delta, B, L, H, dropout_p, seed, stream_id, dqkv, st);
    default: return PTAMD_ERR_BAD_SHAPE;
  }
}

}  // extern "C"
