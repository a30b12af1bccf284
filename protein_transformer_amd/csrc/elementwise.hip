// Bandwidth-bound pieces of the encoder: embedding + positional encoding, LayerNorm, bias-gradient column
// sums and the elementwise backward helpers.  All are HBM-bound: 16-byte accesses, one wavefront per row,
// fixed-order reductions (no float atomics, results are run-to-run deterministic).
//
// Reference semantics:
//   Embeddings * sqrt(D)          /root/reference/protein_transformer/models/transformer/Sublayers.py:65-72
//   PositionalEncoding (+dropout) .../Sublayers.py:37-62 and the second add + dropout of Encoder.py:30
//   torch.nn.LayerNorm(D)         .../Sublayers.py:13,17   (eps 1e-5, biased variance, affine)
//   Dropout / ReLU / tanh backward: autograd of Sublayers.py:17,34 and encoder_only.py:41
#include "common.h"
#include <type_traits>
#include "hp_format.h"

namespace {

constexpr uint32_t STREAM_EMB1 = 0xE1u, STREAM_EMB2 = 0xE2u;

// ------------------------------------------------------------------------------------------------ embedding
__global__ void embed_fwd_kernel(const int64_t *__restrict__ seq, const float *__restrict__ emb,
                                 const float *__restrict__ pe, int L, int D, int64_t n4, float p, uint64_t seed,
                                 float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of one token
  if (i >= n4) return;
  const int d4 = D >> 2;
  const int64_t t = i / d4;
  const int c = (int)(i - t * d4) * 4;
  const int l = (int)(t % L);
  int64_t id = seq[t];
  if (id < 0 || id > 21) id = 21;
  const float sq = sqrtf((float)D);
  const float4 e = *reinterpret_cast<const float4 *>(emb + id * D + c);
  const float4 q = *reinterpret_cast<const float4 *>(pe + (size_t)l * D + c);
  float x0[4] = {e.x * sq, e.y * sq, e.z * sq, e.w * sq};
  const float pv[4] = {q.x, q.y, q.z, q.w};
  float o[4];
  if (p > 0.f) {
    const uint32_t thr = dropout_threshold(p);
    const float ks = 1.f / (1.f - p);
    const uint4 r1 = pt_rand4(seed, (uint64_t)i, STREAM_EMB1), r2 = pt_rand4(seed, (uint64_t)i, STREAM_EMB2);
    const uint32_t w1[4] = {r1.x, r1.y, r1.z, r1.w}, w2[4] = {r2.x, r2.y, r2.z, r2.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float inner = w1[k] >= thr ? (x0[k] + pv[k]) * ks : 0.f;   // PositionalEncoding's dropout
      o[k] = w2[k] >= thr ? (x0[k] + inner) * ks : 0.f;                // Encoder.emb_dropout
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = x0[k] + (x0[k] + pv[k]);
  }
  *reinterpret_cast<float4 *>(out + i * 4) = make_float4(o[0], o[1], o[2], o[3]);
}

// partial embedding-gradient tables: block (slab of 256 columns, chunk of tokens) -> part[chunk][22][D].  A lane owns FOUR
// consecutive columns (round 5): the two generator calls of a (token, column quadruple) serve all four of them - with a
// column per lane, as in rounds 1-4, four lanes drew the same words (40 us for 16384 x 512, all of it the hash: now 256
// chunks of tokens x 2 wavefronts, a quarter of the draws).
constexpr int EMB_CHUNKS = 256, EMB_WAVES = 2;
__global__ __launch_bounds__(64 * EMB_WAVES) void embed_bwd_kernel(const int64_t *__restrict__ seq,
                                                                    const float *__restrict__ dout, int64_t T, int D, float p,
                                                                    uint64_t seed, float *__restrict__ part) {
  __shared__ float4 tab[EMB_WAVES][22][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  for (int v = 0; v < 22; ++v) tab[wave][v][lane] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t per = (T + EMB_CHUNKS - 1) / EMB_CHUNKS;
  const int64_t t0 = blockIdx.y * per, t1 = min(T, t0 + per);
  const float sq = sqrtf((float)D);
  const uint32_t thr = dropout_threshold(p);
  const float ks = 1.f / (1.f - p);
  if (c < D) {
    // EIGHT tokens per round: their ids and gradient rows are requested first, the table updates (a dependent LDS
    // read-modify-write chain) follow - one token per round trip to memory was the kernel's time, with or without dropout
    constexpr int U = 8;
    for (int64_t tb = t0 + wave; tb < t1; tb += EMB_WAVES * U) {
      int64_t id[U];
      float4 g[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t t = min(tb + (int64_t)u * EMB_WAVES, t1 - 1);   // (clamped: a repeated token is skipped below)
        id[u] = seq[t];
        g[u] = *reinterpret_cast<const float4 *>(dout + t * D + c);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t t = tb + (int64_t)u * EMB_WAVES;
        if (t >= t1) break;
        int64_t v = id[u];
        if (v < 0 || v > 21) v = 21;
        float4 x = g[u];
        if (p > 0.f) {
          const uint64_t i4 = (uint64_t)(t * (D >> 2) + (c >> 2));
          const uint4 r1 = pt_rand4(seed, i4, STREAM_EMB1), r2 = pt_rand4(seed, i4, STREAM_EMB2);
          x.x = r2.x >= thr ? x.x * ks * (1.f + (r1.x >= thr ? ks : 0.f)) : 0.f;
          x.y = r2.y >= thr ? x.y * ks * (1.f + (r1.y >= thr ? ks : 0.f)) : 0.f;
          x.z = r2.z >= thr ? x.z * ks * (1.f + (r1.z >= thr ? ks : 0.f)) : 0.f;
          x.w = r2.w >= thr ? x.w * ks * (1.f + (r1.w >= thr ? ks : 0.f)) : 0.f;
        } else {
          x.x *= 2.f; x.y *= 2.f; x.z *= 2.f; x.w *= 2.f;
        }
        float4 a = tab[wave][v][lane];
        a.x += x.x * sq; a.y += x.y * sq; a.z += x.z * sq; a.w += x.w * sq;
        tab[wave][v][lane] = a;
      }
    }
  }
  __syncthreads();
  if (c < D)
    for (int v = wave; v < 22; v += EMB_WAVES) {
      const float4 a = tab[0][v][lane], b = tab[1][v][lane];
      *reinterpret_cast<float4 *>(part + ((size_t)blockIdx.y * 22 + v) * D + c) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}
// demb[v][c] += sum over the chunks, in a fixed order: block = (64 columns, vocabulary row v), four groups of threads take every
// fourth chunk with eight loads in flight each (one thread walking all 256 partials was a chain of 256 dependent-looking loads)
__global__ __launch_bounds__(256) void embed_bwd_reduce_kernel(const float *__restrict__ part, int D, float *__restrict__ demb) {
  __shared__ float s_red[4][64];
  const int cx = threadIdx.x & 63, cy = threadIdx.x >> 6, c = blockIdx.x * 64 + cx, v = blockIdx.y;
  float s = 0.f;
  if (c < D) {
    const float *q = part + (size_t)v * D + c;
    const size_t step = (size_t)22 * D;
    for (int ch = cy; ch < EMB_CHUNKS; ch += 32) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = ch + 4 * u < EMB_CHUNKS ? q[(size_t)(ch + 4 * u) * step] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
  }
  s_red[cy][cx] = s;
  __syncthreads();
  if (cy == 0 && c < D) demb[(size_t)v * D + c] += ((s_red[0][cx] + s_red[1][cx]) + s_red[2][cx]) + s_red[3][cx];
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// one wavefront per row, NV float4 per lane kept in registers: everything behind the loads of the row
template <int NV>
__device__ __forceinline__ void layernorm_fwd_row(float4 (&v)[NV], int64_t row, int lane, int D, const float *__restrict__ gamma,
                                                  const float *__restrict__ beta, float *__restrict__ y,
                                                  float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                                  uint32_t *__restrict__ row_scale, char *__restrict__ planes) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    if (c < D) {
      const float a = v[j].x - mean, b = v[j].y - mean, cc = v[j].z - mean, d = v[j].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
  float *yr = y + row * D;
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    if (c < D) {
      const float4 g = *reinterpret_cast<const float4 *>(gamma + c), b = *reinterpret_cast<const float4 *>(beta + c);
      const float4 o = make_float4((v[j].x - mean) * rstd * g.x + b.x, (v[j].y - mean) * rstd * g.y + b.y,
                                   (v[j].z - mean) * rstd * g.z + b.z, (v[j].w - mean) * rstd * g.w + b.w);
      *reinterpret_cast<float4 *>(yr + c) = o;
      v[j] = o;
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
  }
  if (row_scale) amax = wave_max(amax);  // the f16x2 row scale of y as a GEMM operand (include/ptamd.h), for free here
  const uint32_t sbits = pt_row_scale_bits(__float_as_uint(amax));
  if (planes) {   // y a second time, pre-split for ptamd_gemm_hp (csrc/hp_format.h): the A operand of the product behind
    const float sc = __uint_as_float(sbits);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (j * 64 + lane) * 4;
      if (c < D) pthp::store4_split(planes, D >> 4, row, c, v[j], sc);
    }
  }
  if (lane == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
    if (row_scale) row_scale[row] = sbits;
  }
}

template <int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, int64_t T, int D,
                                                            float *__restrict__ y, float *__restrict__ mean_out,
                                                            float *__restrict__ rstd_out, uint32_t *__restrict__ row_scale,
                                                            char *__restrict__ planes) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  const float *xr = x + row * D;
  float4 v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    v[j] = c < D ? *reinterpret_cast<const float4 *>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  layernorm_fwd_row<NV>(v, row, lane, D, gamma, beta, y, mean_out, rstd_out, row_scale, planes);
}

// Round 6: the LayerNorm behind a split product that was left unreduced (PTAMD_EPI_SLABS) - the rows are first MADE here,
// x = residual + dropout(sum of the NS K slices + bias), in the order and with the decisions of gemm_splitk_reduce_kernel
// (same bits), written to x_out, and normalised from registers.  The four rows of a workgroup (4 k ... 4 k + 3) take
// their dropout decisions from the SAME generator calls (drop_call_index: fields e = row & 3 of one call per column): the
// words are drawn once per workgroup - columns t and t + 256 by thread t - and shared through LDS.  D <= 512.
template <int NV, int NS>
__global__ __launch_bounds__(256) void layernorm_fwd_sum_kernel(
    const float *__restrict__ slabs, int64_t slab, const float *__restrict__ bias, const float *__restrict__ residual,
    float p, uint64_t seed, uint32_t stream_id, float *__restrict__ x_out, const float *__restrict__ gamma,
    const float *__restrict__ beta, int64_t T, int D, float *__restrict__ y, float *__restrict__ mean_out,
    float *__restrict__ rstd_out, uint32_t *__restrict__ row_scale, char *__restrict__ planes) {
  __shared__ uint2 words[NV * 256];   // per column: the two words of the call that hold the fields of rows row0 ... row0 + 3
  const int lane = threadIdx.x & 63, e_row = threadIdx.x >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * 4, row = row0 + e_row;
  if (p > 0.f) {
    const bool upper = (row0 >> 3) & 1;   // drop_field: fields 4 ... 7 (words z, w) for rows 8 ... 15 of every 16
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int col = k * 256 + threadIdx.x;
      if (col < D) {
        const uint4 r = pt_rand4(seed, drop_call_index(row0, col, D), stream_id);
        words[col] = upper ? make_uint2(r.z, r.w) : make_uint2(r.x, r.y);
      }
    }
    __syncthreads();
  }
  if (row >= T) return;
  const uint32_t thr16 = dropout_threshold(p) >> 16;
  const float keep_scale = 1.f / (1.f - p);
  float4 v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < D) {
      float4 part[NS];
#pragma unroll
      for (int k = 0; k < NS; ++k) part[k] = *reinterpret_cast<const float4 *>(slabs + (int64_t)k * slab + row * D + c);
      const float4 bv = *reinterpret_cast<const float4 *>(bias + c);
      const float4 rv = *reinterpret_cast<const float4 *>(residual + row * D + c);
      float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < NS; ++k) { a[0] += part[k].x; a[1] += part[k].y; a[2] += part[k].z; a[3] += part[k].w; }
      const float bb[4] = {bv.x, bv.y, bv.z, bv.w}, rr[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = a[e] + bb[e];
        if (p > 0.f) {
          const uint2 u = words[c + e];
          const uint32_t wd = (e_row >> 1) ? u.y : u.x, fv = (e_row & 1) ? wd >> 16 : wd & 0xffffu;
          t = fv >= thr16 ? t * keep_scale : 0.f;
        }
        a[e] = t + rr[e];
      }
      v[j] = make_float4(a[0], a[1], a[2], a[3]);
      *reinterpret_cast<float4 *>(x_out + row * D + c) = v[j];
    }
  }
  layernorm_fwd_row<NV>(v, row, lane, D, gamma, beta, y, mean_out, rstd_out, row_scale, planes);
}

#ifndef PT_LN_BWD_BLOCKS
#define PT_LN_BWD_BLOCKS 512
#endif
constexpr int LN_BWD_BLOCKS = PT_LN_BWD_BLOCKS;  // 2 per CU, 4 wavefronts each; one partial row of (dgamma, dbeta) per block
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ mean, const float *__restrict__ rstd,
                                                            const float *__restrict__ dres, int64_t T, int D,
                                                            float *__restrict__ dx, float *__restrict__ part) {
  const int lane = threadIdx.x & 63, wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  float4 g[NV], dg[NV], db[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    g[j] = c < D ? *reinterpret_cast<const float4 *>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    dg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int64_t row = wid; row < T; row += LN_BWD_BLOCKS * 4) {
    const float mu = mean[row], rs = rstd[row];
    float4 xh[NV], gy[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (j * 64 + lane) * 4;
      if (c < D) {
        const float4 xv = *reinterpret_cast<const float4 *>(x + row * D + c);
        const float4 d = *reinterpret_cast<const float4 *>(dy + row * D + c);
        xh[j] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        gy[j] = make_float4(d.x * g[j].x, d.y * g[j].y, d.z * g[j].z, d.w * g[j].w);
        s1 += (gy[j].x + gy[j].y) + (gy[j].z + gy[j].w);
        s2 += (gy[j].x * xh[j].x + gy[j].y * xh[j].y) + (gy[j].z * xh[j].z + gy[j].w * xh[j].w);
        dg[j].x += d.x * xh[j].x; dg[j].y += d.y * xh[j].y; dg[j].z += d.z * xh[j].z; dg[j].w += d.w * xh[j].w;
        db[j].x += d.x; db[j].y += d.y; db[j].z += d.z; db[j].w += d.w;
      }
    }
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (j * 64 + lane) * 4;
      if (c < D) {
        float4 o = make_float4(rs * (gy[j].x - m1 - xh[j].x * m2), rs * (gy[j].y - m1 - xh[j].y * m2),
                               rs * (gy[j].z - m1 - xh[j].z * m2), rs * (gy[j].w - m1 - xh[j].w * m2));
        if (dres) {  // gradient arriving through the residual connection around the normalised sublayer
          const float4 r = *reinterpret_cast<const float4 *>(dres + row * D + c);
          o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        *reinterpret_cast<float4 *>(dx + row * D + c) = o;
      }
    }
  }
  // the four wavefronts of the block add up their (dgamma, dbeta) in a fixed order: one partial row per block
  __shared__ float4 s_dg[3][NV * 64], s_db[3][NV * 64];
  const int wave = threadIdx.x >> 6;
  if (wave > 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      s_dg[wave - 1][j * 64 + lane] = dg[j];
      s_db[wave - 1][j * 64 + lane] = db[j];
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (j * 64 + lane) * 4;
      if (c < D) {
        float4 a = dg[j], b = db[j];
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          const float4 pa = s_dg[w][j * 64 + lane], pb = s_db[w][j * 64 + lane];
          a.x += pa.x; a.y += pa.y; a.z += pa.z; a.w += pa.w;
          b.x += pb.x; b.y += pb.y; b.z += pb.z; b.w += pb.w;
        }
        *reinterpret_cast<float4 *>(part + ((size_t)blockIdx.x * 2) * D + c) = a;
        *reinterpret_cast<float4 *>(part + ((size_t)blockIdx.x * 2 + 1) * D + c) = b;
      }
    }
  }
}
// LayerNorm backward FUSED with the dropout backward that always follows it in the encoder (the output dx is the gradient
// of the residual stream; the next sublayer's GEMMs take dropout'(dx)), and with the f16x2 bookkeeping of that operand:
//   dx       = LN'(dy) + dres                                            (as layernorm_bwd_kernel)
//   dropped  = dx * mask / (1 - p), mask of the GEMM epilogue that drew it (common.h: drop_call_index / drop_field)
//   row_scale[t]   = f16x2 scale of row t of `dropped` (exact row maximum)
//   bound_scale[t] = f16x2 scale for a row bounded by |dropped[t]|_2 * *bound_factor - the rows of the product
//                    dropped W that the next GEMM writes (Cauchy-Schwarz with the largest column norm of W)
// One generator call serves the 8 rows {r0, r0+1, r0+2, r0+3, r0+8, .., r0+11} of a column, so a wavefront takes such a
// GROUP of rows, draws its words once (4 NV calls per lane) and walks the 8 rows with them.
// HALVES = 2 or 4 (few tokens: fewer 8-row groups than wavefronts): a group is shared by two (four) wavefronts - 4 (2)
// consecutive rows of the group's row order each - which all draw the group's words: a half (quarter) of the serial row
// chain per wavefront, which is what a launch with idle wavefronts is bound by (2048 tokens: 21.6 us with one wavefront
// per group, 16.3 with two).
// NS > 1 (round 6): dy = the sum of NS slabs `slab` floats apart (the unreduced K slices of the split product that made it),
// added in slab order behind their loads: (((s0 + s1) + s2) + s3), the bits of gemm_splitk_reduce_plain_kernel.
template <int NV, int HALVES, int NS = 1>
__global__ __launch_bounds__(256) void layernorm_bwd_dropout_kernel(
    const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ dres, int64_t T, int D, float *__restrict__ dx,
    float *__restrict__ part, float p, uint64_t seed, uint32_t stream_id, float *__restrict__ dropped,
    uint32_t *__restrict__ row_scale, const float *__restrict__ bound_factor, uint32_t *__restrict__ bound_scale,
    uint32_t *__restrict__ row_scale_min, uint32_t *__restrict__ bound_scale_min, char *__restrict__ planes, int64_t slab = 0) {
  const int lane = threadIdx.x & 63, wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  uint32_t rmin = 0x7F000000u, bmin = 0x7F000000u;  // smallest scale = largest row seen by this wavefront
  float4 g[NV], dg[NV], db[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    g[j] = c < D ? *reinterpret_cast<const float4 *>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    dg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const uint32_t t16 = dropout_threshold(p) >> 16;
  const float ks = 1.f / (1.f - p), bf = bound_factor ? *bound_factor : 0.f;
  const float *__restrict__ rsrc = dres ? dres : dy;   // no residual gradient: the loads stay (unconditional), their weight is 0
  const float rflag = dres ? 1.f : 0.f;
  const int64_t ngroups = ((T + 31) / 32) * 4;
  for (int64_t item = wid; item < ngroups * HALVES; item += LN_BWD_BLOCKS * 4) {
    const int64_t cr = item / HALVES;  // cr = call row (4 I + 2 h + gp)
    const int half = (int)(item % HALVES);  // (wavefront-uniform)
    const int64_t r0 = ((cr >> 2) << 5) + 16 * (cr & 1) + 4 * ((cr >> 1) & 1);
    uint4 rnd[NV][4];
    if (p > 0.f) {
#pragma unroll
      for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = (j * 64 + lane) * 4 + k;
          rnd[j][k] = pt_rand4(seed, (uint64_t)cr * (uint64_t)D + (uint64_t)(c < D ? c : 0), stream_id);
        }
    }
    // The eight rows of the group, software-pipelined: the loads of row f + 1 (x, dy, dres, mean, rstd: UNCONDITIONAL, the
    // row index clamped, columns beyond D clamped and masked afterwards) are issued before row f is reduced and written.
    // Written with predicated loads the compiler turned every `if` into a branch, issued the loads two at a time and
    // waited for each pair: four dependent memory round trips per row, 3.3 TB/s (profiles/r03).
    float4 xa[2][NV], da[2][NV], ra[2][NV];
    float4 ds[2][NS > 1 ? NS - 1 : 1][NV];   // the later slabs of dy (NS > 1)
    float mua[2], rsa[2];
    auto fetch = [&](int f, int set) __attribute__((always_inline)) {
      const int64_t row = min(r0 + 8 * (f >> 2) + (f & 3), T - 1);
      mua[set] = mean[row];
      rsa[set] = rstd[row];
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int c = min((j * 64 + lane) * 4, D - 4);
        xa[set][j] = *reinterpret_cast<const float4 *>(x + row * D + c);
        da[set][j] = *reinterpret_cast<const float4 *>(dy + row * D + c);
        ra[set][j] = *reinterpret_cast<const float4 *>(rsrc + row * D + c);
#pragma unroll
        for (int k = 1; k < NS; ++k) ds[set][k - 1][j] = *reinterpret_cast<const float4 *>(dy + (int64_t)k * slab + row * D + c);
      }
    };
    auto rows_from = [&](auto first) __attribute__((always_inline)) {
    constexpr int F0 = decltype(first)::value, F1 = F0 + 8 / HALVES;
    fetch(F0, F0 & 1);
#pragma unroll
    for (int f = F0; f < F1; ++f) {
      const int64_t row = r0 + 8 * (f >> 2) + (f & 3);
      if (f < F1 - 1) fetch(f + 1, (f + 1) & 1);
      if (row >= T) continue;  // wavefront-uniform (no loads behind it)
      const float mu = mua[f & 1], rs = rsa[f & 1];
      float4 xh[NV], gy[NV];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int c = (j * 64 + lane) * 4;
        const float live = c < D ? 1.f : 0.f;   // lanes beyond D hold a clamped copy of the last columns: weight 0
        const float4 xv = xa[f & 1][j];
        float4 dsum = da[f & 1][j];
#pragma unroll
        for (int k = 1; k < NS; ++k) {   // slab order: the reduction launch's sum
          const float4 o = ds[f & 1][k - 1][j];
          dsum.x += o.x; dsum.y += o.y; dsum.z += o.z; dsum.w += o.w;
        }
        const float4 d = make_float4(dsum.x * live, dsum.y * live, dsum.z * live, dsum.w * live);
        xh[j] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        gy[j] = make_float4(d.x * g[j].x, d.y * g[j].y, d.z * g[j].z, d.w * g[j].w);
        s1 += (gy[j].x + gy[j].y) + (gy[j].z + gy[j].w);
        s2 += (gy[j].x * xh[j].x + gy[j].y * xh[j].y) + (gy[j].z * xh[j].z + gy[j].w * xh[j].w);
        dg[j].x += d.x * xh[j].x; dg[j].y += d.y * xh[j].y; dg[j].z += d.z * xh[j].z; dg[j].w += d.w * xh[j].w;
        db[j].x += d.x; db[j].y += d.y; db[j].z += d.z; db[j].w += d.w;
      }
      const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
      float amax = 0.f, sq = 0.f;
      float4 od[NV];   // the dropped row, kept for the plane output below (its scale is known only after the whole row)
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int c = (j * 64 + lane) * 4;
        od[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < D) {
          // rs * (..) + r as ONE fma each (r * rflag is exact: rflag is 0 or 1, no residual gradient -> + 0).  The unfused
          // layernorm_bwd_kernel contracts differently, so the two agree to rounding, not bit for bit
          // (tests/test_gpu_scales.py compares them at 2e-6 relative)
          const float4 r = ra[f & 1][j];
          float4 o = make_float4(fmaf(rs, gy[j].x - m1 - xh[j].x * m2, r.x * rflag), fmaf(rs, gy[j].y - m1 - xh[j].y * m2, r.y * rflag),
                                 fmaf(rs, gy[j].z - m1 - xh[j].z * m2, r.z * rflag), fmaf(rs, gy[j].w - m1 - xh[j].w * m2, r.w * rflag));
          *reinterpret_cast<float4 *>(dx + row * D + c) = o;
          if (p > 0.f) {
            o.x = drop_field_value(rnd[j][0], f) >= t16 ? o.x * ks : 0.f;
            o.y = drop_field_value(rnd[j][1], f) >= t16 ? o.y * ks : 0.f;
            o.z = drop_field_value(rnd[j][2], f) >= t16 ? o.z * ks : 0.f;
            o.w = drop_field_value(rnd[j][3], f) >= t16 ? o.w * ks : 0.f;
            *reinterpret_cast<float4 *>(dropped + row * D + c) = o;
          }
          amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
          sq += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
          od[j] = o;
        }
      }
      amax = wave_max(amax);
      sq = wave_sum(sq);
      const uint32_t rs_bits = pt_row_scale_bits(__float_as_uint(amax)), bs_bits = pt_row_scale_bits(__float_as_uint(sqrtf(sq) * bf));
      if (planes) {   // the dropped row a second time, pre-split for ptamd_gemm_hp (csrc/hp_format.h): the A operand of the dX
                      // product behind it reads these by LDS-DMA instead of splitting the fp32 copy while it stages it
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const int c = (j * 64 + lane) * 4;
          if (c < D) pthp::store4_split(planes, D >> 4, row, c, od[j], __uint_as_float(rs_bits));
        }
      }
      rmin = min(rmin, rs_bits);
      bmin = min(bmin, bs_bits);
      if (lane == 0) {
        if (row_scale) row_scale[row] = rs_bits;
        if (bound_scale) bound_scale[row] = bs_bits;
      }
    }
    };
    // part `half` of the group's eight rows: rows half (8 / HALVES) .. of its row order
#define PT_LN_PART(h)                                                                     \
  case h:                                                                                 \
    if constexpr ((h) < HALVES) rows_from(std::integral_constant<int, (h) * (8 / HALVES)>{}); \
    break;
    switch (half) {
      PT_LN_PART(0) PT_LN_PART(1) PT_LN_PART(2) PT_LN_PART(3) PT_LN_PART(4) PT_LN_PART(5) PT_LN_PART(6) PT_LN_PART(7)
      default: break;
    }
#undef PT_LN_PART
  }
  __shared__ float4 s_dg[3][NV * 64], s_db[3][NV * 64];
  __shared__ uint32_t s_min[2][4];
  const int wave = threadIdx.x >> 6;
  if (row_scale_min || bound_scale_min) {  // one atomic per block and copy (atomics on one address serialise in the L2)
    if (lane == 0) {
      s_min[0][wave] = rmin;
      s_min[1][wave] = bmin;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
      const uint32_t a = min(min(s_min[0][0], s_min[0][1]), min(s_min[0][2], s_min[0][3]));
      const uint32_t c = min(min(s_min[1][0], s_min[1][1]), min(s_min[1][2], s_min[1][3]));
      if (row_scale_min) atomicMin(row_scale_min + threadIdx.x, a);
      if (bound_scale_min && bound_factor) atomicMin(bound_scale_min + threadIdx.x, c);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      s_dg[wave - 1][j * 64 + lane] = dg[j];
      s_db[wave - 1][j * 64 + lane] = db[j];
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (j * 64 + lane) * 4;
      if (c < D) {
        float4 a = dg[j], b = db[j];
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          const float4 pa = s_dg[w][j * 64 + lane], pb = s_db[w][j * 64 + lane];
          a.x += pa.x; a.y += pa.y; a.z += pa.z; a.w += pa.w;
          b.x += pb.x; b.y += pb.y; b.z += pb.z; b.w += pb.w;
        }
        *reinterpret_cast<float4 *>(part + ((size_t)blockIdx.x * 2) * D + c) = a;
        *reinterpret_cast<float4 *>(part + ((size_t)blockIdx.x * 2 + 1) * D + c) = b;
      }
    }
  }
}
// LN_BWD_BLOCKS partial rows -> [D]: one block per 64 columns, 16 waves each summing every 16th row (256 B reads).
// Several LayerNorm sites per launch (blockIdx.y = site): the backward kernels of a layer (or of the whole pass) leave
// their partial rows in separate workspaces and ONE reduce launch finishes them (ptamd_layernorm_bwd_reduce).
constexpr int LN_MAX_REDUCE_JOBS = 16;
struct LnReduceJobs {
  const float *part[LN_MAX_REDUCE_JOBS];
  float *dgamma[LN_MAX_REDUCE_JOBS], *dbeta[LN_MAX_REDUCE_JOBS];
  int D[LN_MAX_REDUCE_JOBS];
};
__global__ __launch_bounds__(1024) void layernorm_bwd_reduce_kernel(const LnReduceJobs jobs) {
  __shared__ float s_a[16][64], s_b[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, D = jobs.D[blockIdx.y];
  const float *__restrict__ part = jobs.part[blockIdx.y];
  if ((int)blockIdx.x * 64 >= D) return;
  float a = 0.f, b = 0.f;
  if (c < D)
    for (int w = wave; w < LN_BWD_BLOCKS; w += 16) {
      a += part[((size_t)w * 2) * D + c];
      b += part[((size_t)w * 2 + 1) * D + c];
    }
  s_a[wave][lane] = a;
  s_b[wave][lane] = b;
  __syncthreads();
  if (wave == 0 && c < D) {
    float ta = 0.f, tb = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      ta += s_a[w][lane];
      tb += s_b[w][lane];
    }
    jobs.dgamma[blockIdx.y][c] += ta;
    jobs.dbeta[blockIdx.y][c] += tb;
  }
}
int launch_ln_reduce(const float *part, int D, float *dgamma, float *dbeta, hipStream_t st) {
  LnReduceJobs j;
  for (int i = 0; i < LN_MAX_REDUCE_JOBS; ++i) { j.part[i] = part; j.dgamma[i] = dgamma; j.dbeta[i] = dbeta; j.D[i] = D; }
  hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3((D + 63) / 64, 1), dim3(1024), 0, st, j);
  return pt_check_launch();
}

// ------------------------------------------------------------------------------------------------ column sums
constexpr int CS_CHUNKS = 128;
__global__ void colsum_partial_kernel(const float *__restrict__ x, int64_t T, int N, int ldx, float *__restrict__ part) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  const int64_t per = (T + CS_CHUNKS - 1) / CS_CHUNKS;
  const int64_t t0 = blockIdx.y * per, t1 = min(T, t0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int64_t t = t0;
  for (; t + 3 < t1; t += 4) {
    s0 += x[t * ldx + c];
    s1 += x[(t + 1) * ldx + c];
    s2 += x[(t + 2) * ldx + c];
    s3 += x[(t + 3) * ldx + c];
  }
  for (; t < t1; ++t) s0 += x[t * ldx + c];
  part[(size_t)blockIdx.y * N + c] = (s0 + s1) + (s2 + s3);
}
__global__ void colsum_reduce_kernel(const float *__restrict__ part, int N, int accumulate, float *__restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  float s = 0.f;
  for (int ch = 0; ch < CS_CHUNKS; ++ch) s += part[(size_t)ch * N + c];
  out[c] = accumulate ? out[c] + s : s;
}

// ------------------------------------------------------------------------------------------------ elementwise bwd
__global__ void relu_dropout_bwd_kernel(const float4 *__restrict__ dy, const float4 *__restrict__ y, int64_t n4,
                                        float ks, float4 *__restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 d = dy[i], v = y[i];
  dx[i] = make_float4(v.x > 0.f ? d.x * ks : 0.f, v.y > 0.f ? d.y * ks : 0.f, v.z > 0.f ? d.z * ks : 0.f,
                      v.w > 0.f ? d.w * ks : 0.f);
}
__global__ void tanh_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y, int64_t n,
                                float *__restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dx[i] = dy[i] * (1.f - y[i] * y[i]);
}
// the GEMM epilogue's mask (common.h: drop_call_index / drop_field): one thread = one call = 8 rows of one column
__global__ void dropout_bwd_kernel(const float *__restrict__ dy, int64_t rows, int cols, float p, uint64_t seed,
                                   uint32_t stream_id, float *__restrict__ dx) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t cr = blockIdx.y;  // call row: (I, h, g >> 1) = (cr >> 2, (cr >> 1) & 1, cr & 1)
  if (c >= cols) return;
  const uint32_t t16 = dropout_threshold(p) >> 16;
  const float ks = 1.f / (1.f - p);
  const uint4 r = pt_rand4(seed, (uint64_t)cr * cols + c, stream_id);
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    const int64_t row = ((cr >> 2) << 5) + 8 * (2 * (cr & 1) + (f >> 2)) + 4 * ((cr >> 1) & 1) + (f & 3);
    if (row < rows) dx[row * cols + c] = drop_field_value(r, f) >= t16 ? dy[row * cols + c] * ks : 0.f;
  }
}

}  // namespace

extern "C" {

int ptamd_embed_fwd(const int64_t *seq, const float *emb, const float *pe, int B, int L, int D, float dropout_p,
                    uint64_t seed, float *out, void *stream) {
  if (B <= 0 || L <= 0 || D <= 0 || (D & 3)) return PTAMD_ERR_BAD_SHAPE;
  const int64_t n4 = (int64_t)B * L * (D >> 2);
  hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seq, emb,
                     pe, L, D, n4, dropout_p, seed, out);
  return pt_check_launch();
}

size_t ptamd_embed_bwd_workspace_bytes(int D) { return (size_t)EMB_CHUNKS * 22 * (D > 0 ? D : 0) * sizeof(float); }

int ptamd_embed_bwd(const int64_t *seq, const float *dout, int B, int L, int D, float dropout_p, uint64_t seed,
                    float *demb, void *workspace, size_t workspace_bytes, void *stream) {
  if (B <= 0 || L <= 0 || D <= 0 || (D & 3)) return PTAMD_ERR_BAD_SHAPE;
  if (!workspace || workspace_bytes < ptamd_embed_bwd_workspace_bytes(D)) return PTAMD_ERR_WORKSPACE;
  float *part = static_cast<float *>(workspace);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((D + 255) / 256, EMB_CHUNKS), dim3(64 * EMB_WAVES), 0, st, seq, dout, (int64_t)B * L, D,
                     dropout_p, seed, part);
  int rc = pt_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(embed_bwd_reduce_kernel, dim3((D + 63) / 64, 22), dim3(256), 0, st, part, D, demb);
  return pt_check_launch();
}

int ptamd_layernorm_fwd(const float *x, const float *gamma, const float *beta, int64_t T, int D, float *y, float *mean,
                        float *rstd, uint32_t *row_scale, void *planes_out, void *stream) {
  if (T <= 0 || D <= 0 || (D & 3) || D > 2048) return PTAMD_ERR_BAD_SHAPE;
  if (planes_out && (!row_scale || (D & 31) || !pt_aligned16(planes_out))) return PTAMD_ERR_BAD_SHAPE;
  char *planes = static_cast<char *>(planes_out);
  const dim3 grid((unsigned)((T + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (D <= 256) hipLaunchKernelGGL(layernorm_fwd_kernel<1>, grid, block, 0, st, x, gamma, beta, T, D, y, mean, rstd, row_scale, planes);
  else if (D <= 512) hipLaunchKernelGGL(layernorm_fwd_kernel<2>, grid, block, 0, st, x, gamma, beta, T, D, y, mean, rstd, row_scale, planes);
  else if (D <= 1024) hipLaunchKernelGGL(layernorm_fwd_kernel<4>, grid, block, 0, st, x, gamma, beta, T, D, y, mean, rstd, row_scale, planes);
  else hipLaunchKernelGGL(layernorm_fwd_kernel<8>, grid, block, 0, st, x, gamma, beta, T, D, y, mean, rstd, row_scale, planes);
  return pt_check_launch();
}

int ptamd_layernorm_fwd_sum(const float *slabs, int n_slabs, int64_t slab_stride, const float *bias, const float *residual,
                            float dropout_p, uint64_t seed, uint32_t stream_id, float *x_out, const float *gamma,
                            const float *beta, int64_t T, int D, float *y, float *mean, float *rstd, uint32_t *row_scale,
                            void *planes_out, void *stream) {
  if (T <= 0 || D <= 0 || (D & 3) || D > 512) return PTAMD_ERR_BAD_SHAPE;
  if (n_slabs < 2 || n_slabs > 4 || slab_stride < T * (int64_t)D || (slab_stride & 3)) return PTAMD_ERR_BAD_SHAPE;
  if (!slabs || !bias || !residual || !x_out || dropout_p < 0.f || dropout_p >= 1.f) return PTAMD_ERR_BAD_SHAPE;
  if (!pt_aligned16(slabs) || !pt_aligned16(bias) || !pt_aligned16(residual) || !pt_aligned16(x_out)) return PTAMD_ERR_ALIGN;
  if (planes_out && (!row_scale || (D & 31) || !pt_aligned16(planes_out))) return PTAMD_ERR_BAD_SHAPE;
  char *planes = static_cast<char *>(planes_out);
  const dim3 grid((unsigned)((T + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define PT_LN_SUM(NV, NS)                                                                                                      \
  hipLaunchKernelGGL((layernorm_fwd_sum_kernel<NV, NS>), grid, block, 0, st, slabs, slab_stride, bias, residual, dropout_p, seed, \
                     stream_id, x_out, gamma, beta, T, D, y, mean, rstd, row_scale, planes)
#define PT_LN_SUM_BY_N(NV) do { if (n_slabs == 2) PT_LN_SUM(NV, 2); else if (n_slabs == 3) PT_LN_SUM(NV, 3); else PT_LN_SUM(NV, 4); } while (0)
  if (D <= 256) PT_LN_SUM_BY_N(1);
  else PT_LN_SUM_BY_N(2);
#undef PT_LN_SUM_BY_N
#undef PT_LN_SUM
  return pt_check_launch();
}

size_t ptamd_layernorm_bwd_workspace_bytes(int D) {
  return (size_t)LN_BWD_BLOCKS * 2 * (D > 0 ? D : 0) * sizeof(float);
}

int ptamd_layernorm_bwd(const float *dy, const float *x, const float *gamma, const float *mean, const float *rstd,
                        const float *dres, int64_t T, int D, float *dx, float *dgamma, float *dbeta, void *workspace,
                        size_t workspace_bytes, void *stream) {
  if (T <= 0 || D <= 0 || (D & 3) || D > 2048) return PTAMD_ERR_BAD_SHAPE;
  if (!workspace || workspace_bytes < ptamd_layernorm_bwd_workspace_bytes(D)) return PTAMD_ERR_WORKSPACE;
  float *part = static_cast<float *>(workspace);
  const dim3 grid(LN_BWD_BLOCKS), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (D <= 256) hipLaunchKernelGGL(layernorm_bwd_kernel<1>, grid, block, 0, st, dy, x, gamma, mean, rstd, dres, T, D, dx, part);
  else if (D <= 512) hipLaunchKernelGGL(layernorm_bwd_kernel<2>, grid, block, 0, st, dy, x, gamma, mean, rstd, dres, T, D, dx, part);
  else if (D <= 1024) hipLaunchKernelGGL(layernorm_bwd_kernel<4>, grid, block, 0, st, dy, x, gamma, mean, rstd, dres, T, D, dx, part);
  else hipLaunchKernelGGL(layernorm_bwd_kernel<8>, grid, block, 0, st, dy, x, gamma, mean, rstd, dres, T, D, dx, part);
  int rc = pt_check_launch();
  if (rc || (!dgamma && !dbeta)) return rc;   // no destinations: the partial rows stay in the workspace (ptamd_layernorm_bwd_reduce)
  if (!dgamma || !dbeta) return PTAMD_ERR_BAD_SHAPE;
  return launch_ln_reduce(part, D, dgamma, dbeta, st);
}

int ptamd_layernorm_bwd_dropout(const float *dy, const float *x, const float *gamma, const float *mean, const float *rstd,
                                const float *dres, int64_t T, int D, float dropout_p, uint64_t seed, uint32_t stream_id,
                                float *dx, float *dropped, uint32_t *row_scale, const float *bound_factor,
                                uint32_t *bound_scale, uint32_t *row_scale_min, uint32_t *bound_scale_min,
                                void *dropped_planes, float *dgamma,
                                float *dbeta, int dy_slabs, int64_t dy_slab_stride, void *workspace, size_t workspace_bytes,
                                void *stream) {
  if (T <= 0 || D <= 0 || (D & 3) || D > 1024) return PTAMD_ERR_BAD_SHAPE;  // 4 D / 256 generator words per lane stay in registers
  if (dy_slabs < 1) dy_slabs = 1;
  if (dy_slabs > 4 || (dy_slabs > 1 && (D > 512 || dy_slab_stride < T * (int64_t)D || (dy_slab_stride & 3)))) return PTAMD_ERR_BAD_SHAPE;
  if (dropped_planes && ((D & 31) || !pt_aligned16(dropped_planes))) return PTAMD_ERR_BAD_SHAPE;
  char *planes = static_cast<char *>(dropped_planes);
  if (dropout_p < 0.f || dropout_p >= 1.f || (dropout_p > 0.f && !dropped)) return PTAMD_ERR_BAD_SHAPE;
  if ((bound_scale || bound_scale_min) && !bound_factor) return PTAMD_ERR_BAD_SHAPE;
  if (!workspace || workspace_bytes < ptamd_layernorm_bwd_workspace_bytes(D)) return PTAMD_ERR_WORKSPACE;
  float *part = static_cast<float *>(workspace);
  const dim3 grid(LN_BWD_BLOCKS), block(256);
  hipStream_t st = (hipStream_t)stream;
  // two or four wavefronts per 8-row group while that still leaves every part of a group a wavefront of its own
  const int64_t ngroups = ((T + 31) / 32) * 4, nwaves = (int64_t)LN_BWD_BLOCKS * 4;
  // (one row per wavefront - eight wavefronts per group - measured SLOWER at 2048 tokens: 16.1 against 13.7 us, every
  // wavefront draws the group's words and nothing is prefetched: round 6)
  const int halves = ngroups * 4 <= nwaves ? 4 : ngroups * 2 <= nwaves ? 2 : 1;
#define PT_LN_FUSED_NS(NV, HV, NS)                                                                                                  \
  hipLaunchKernelGGL((layernorm_bwd_dropout_kernel<NV, HV, NS>), grid, block, 0, st, dy, x, gamma, mean, rstd, dres, T, D, dx, part, \
                     dropout_p, seed, stream_id, dropped, row_scale, bound_factor, bound_scale, row_scale_min, bound_scale_min, planes, \
                     dy_slab_stride)
#define PT_LN_FUSED(NV, HV)                                                        \
  do {                                                                             \
    if (dy_slabs == 1) PT_LN_FUSED_NS(NV, HV, 1);                                   \
    else if (NV <= 2 && dy_slabs == 2) PT_LN_FUSED_NS((NV <= 2 ? NV : 1), HV, 2);   \
    else if (NV <= 2 && dy_slabs == 3) PT_LN_FUSED_NS((NV <= 2 ? NV : 1), HV, 3);   \
    else if (NV <= 2) PT_LN_FUSED_NS((NV <= 2 ? NV : 1), HV, 4);                    \
  } while (0)
#define PT_LN_FUSED_BY_T(NV) do { if (halves == 4) PT_LN_FUSED(NV, 4); else if (halves == 2) PT_LN_FUSED(NV, 2); else PT_LN_FUSED(NV, 1); } while (0)
  if (D <= 256) PT_LN_FUSED_BY_T(1);
  else if (D <= 512) PT_LN_FUSED_BY_T(2);
  else PT_LN_FUSED_BY_T(4);
#undef PT_LN_FUSED_BY_T
#undef PT_LN_FUSED
#undef PT_LN_FUSED_NS
  int rc = pt_check_launch();
  if (rc || (!dgamma && !dbeta)) return rc;
  if (!dgamma || !dbeta) return PTAMD_ERR_BAD_SHAPE;
  return launch_ln_reduce(part, D, dgamma, dbeta, st);
}

int ptamd_layernorm_bwd_reduce(const ptamd_ln_reduce_job *jobs, int njobs, void *stream) {
  if (!jobs || njobs <= 0 || njobs > LN_MAX_REDUCE_JOBS) return PTAMD_ERR_BAD_SHAPE;
  LnReduceJobs j;
  int dmax = 0;
  for (int i = 0; i < njobs; ++i) {
    if (!jobs[i].partials || !jobs[i].dgamma || !jobs[i].dbeta || jobs[i].D <= 0) return PTAMD_ERR_BAD_SHAPE;
    j.part[i] = static_cast<const float *>(jobs[i].partials); j.dgamma[i] = jobs[i].dgamma; j.dbeta[i] = jobs[i].dbeta;
    j.D[i] = jobs[i].D;
    dmax = jobs[i].D > dmax ? jobs[i].D : dmax;
  }
  for (int i = njobs; i < LN_MAX_REDUCE_JOBS; ++i) { j.part[i] = j.part[0]; j.dgamma[i] = j.dgamma[0]; j.dbeta[i] = j.dbeta[0]; j.D[i] = 0; }
  hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3((dmax + 63) / 64, njobs), dim3(1024), 0, (hipStream_t)stream, j);
  return pt_check_launch();
}

size_t ptamd_colsum_workspace_bytes(int N) { return (size_t)CS_CHUNKS * (N > 0 ? N : 0) * sizeof(float); }

int ptamd_colsum(const float *x, int64_t T, int N, int ldx, int accumulate, float *out, void *workspace,
                 size_t workspace_bytes, void *stream) {
  if (T <= 0 || N <= 0 || ldx < N) return PTAMD_ERR_BAD_SHAPE;
  if (!workspace || workspace_bytes < ptamd_colsum_workspace_bytes(N)) return PTAMD_ERR_WORKSPACE;
  float *part = static_cast<float *>(workspace);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 255) / 256, CS_CHUNKS), dim3(256), 0, st, x, T, N, ldx, part);
  int rc = pt_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((N + 255) / 256), dim3(256), 0, st, part, N, accumulate, out);
  return pt_check_launch();
}

int ptamd_relu_dropout_bwd(const float *dy, const float *y, int64_t n, float dropout_p, float *dx, void *stream) {
  if (n <= 0 || (n & 3)) return PTAMD_ERR_BAD_SHAPE;
  if (!pt_aligned16(dy) || !pt_aligned16(y) || !pt_aligned16(dx)) return PTAMD_ERR_ALIGN;
  hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4 *>(dy), reinterpret_cast<const float4 *>(y), n / 4,
                     1.f / (1.f - dropout_p), reinterpret_cast<float4 *>(dx));
  return pt_check_launch();
}

int ptamd_tanh_bwd(const float *dy, const float *y, int64_t n, float *dx, void *stream) {
  if (n <= 0) return PTAMD_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, y, n, dx);
  return pt_check_launch();
}

int ptamd_dropout_bwd(const float *dy, int64_t rows, int cols, float dropout_p, uint64_t seed, uint32_t stream_id,
                      float *dx, void *stream) {
  if (rows <= 0 || cols <= 0) return PTAMD_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3((cols + 255) / 256, (unsigned)(((rows + 31) / 32) * 4)), dim3(256), 0,
                     (hipStream_t)stream, dy, rows, cols, dropout_p, seed, stream_id, dx);
  return pt_check_launch();
}

}  // extern "C"
