// ptamd_weights_prep / ptamd_sgd_step_prep / ptamd_adam_step_prep: everything the f16x2 arithmetic wants to know about the
// WEIGHTS of a step, in ONE pass over the flat parameter buffer - optionally the pass of the optimizer that writes them.
//
// Reference: the optimizer step of train.py:41-46,371-381 (clip + SGD / Adam on every parameter) - and, on this side only,
// the bookkeeping of the PTAMD_GEMM_F16X2 arithmetic that rounds 2-4 ran as separate launches in front of every forward
// pass: ptamd_weight_scales (row / column scales and statistics of every weight matrix: two passes over all 76 MB of
// weights, 3.7 launches), ptamd_bound_scales, ptamd_hp_split_rows (W_qkv, W_1 as pre-split planes), ptamd_hp_split_cols
// (W_2^T planes).  Weights change in exactly one place, the optimizer step, which reads and writes every one of them
// anyway; so the step's kernel also leaves behind what the NEXT forward pass needs:
//
//   kernel A  (rows)   one workgroup per 32 rows of a matrix (a wavefront per row, the row in registers): [update,] row
//                      maximum -> row scale, row norm / maximum -> statistics, the row once more as hp planes (hp_format.h) where
//                      asked, column maxima by atomicMax on the bit patterns (order-independent: deterministic), column
//                      sums of squares as fixed-order partials per 32-row block; everything that is not a listed matrix
//                      (embedding, biases, output layer, conv weights) goes through plain elementwise workgroups;
//   kernel B  (columns) column scales from the maxima, W_2^T planes with those scales (operand rows = columns of W_2), the
//                      column norms from the partials (fixed order) and the weight-derived bounds (ptamd_bound_scales'
//                      formula) per layer.
//
// Two launches instead of 3.7 + 1 + 1 + 1 (+ the optimizer's), one read of the weights instead of three, and bit-identical
// scales: the row statistics use the summation order of wscale_kernel, maxima do not depend on an order.  Statistics and
// column maxima are accumulated by atomicMax into buffers that must be zero: they exist twice, kernel B of a call zeroes the
// copy the NEXT call accumulates into (`parity` alternates; the caller zeroes both once).
//
// The tables (segments, block lists, bound jobs) live in DEVICE memory, built once per model by the caller in the layouts of
// include/ptamd.h; the entry points take pointers and counts only.
#include "common.h"
#include "hp_format.h"
#include "optim_update.h"
#include "split_bf16.h"

namespace {

constexpr int RB = 32, WAVES = 4, RW = RB / WAVES, MAXV = 2;   // 32 rows per workgroup, 8 per wavefront; cols <= 64 * 4 * MAXV = 512:
                                                                // wider matrices come as column PANELS of 512 (ptamd_wprep_seg::ld / rowmax_index)
constexpr int PLAIN_F4 = 4096;                                  // float4 per plain workgroup (64 KiB of parameters)

struct OptArgs {
  int kind;                // 0: no update, 1: SGD, 2: Adam
  int zero_g;              // the gradient is zeroed behind its read (the next step's zero_grad, folded in)
  float *g;
  float *m, *v;
  const float *sqnorm;
  float max_norm, lr, wd, beta1, beta2, eps, step_size, inv_sqrt_bc2;
};

using ptopt::clip_coef;
using ptopt::sgd_update;
__device__ __forceinline__ float adam_update(float p, float g, float &m, float &v, float coef, const OptArgs &o) {
  return ptopt::adam_update(p, g, m, v, coef, o.wd, o.beta1, o.beta2, o.eps, o.step_size, o.inv_sqrt_bc2);
}
// the float4 of an update: loaded first (all requests of a row - or of two rows - in flight), applied and stored afterwards
template <int KIND>
struct Quad {
  float4 p, d, m, v;
  __device__ __forceinline__ void load(const float *__restrict__ w, int64_t i, const OptArgs &o) {
    p = *reinterpret_cast<const float4 *>(w + i);
    if (KIND >= 1) d = *reinterpret_cast<const float4 *>(o.g + i);
    if (KIND == 2) {
      m = *reinterpret_cast<const float4 *>(o.m + i);
      v = *reinterpret_cast<const float4 *>(o.v + i);
    }
  }
  __device__ __forceinline__ float4 apply(float *__restrict__ w, int64_t i, const OptArgs &o, float coef) {
    if (KIND == 0) return p;
    if (KIND == 1) {
      p.x = sgd_update(p.x, d.x, coef, o.lr, o.wd); p.y = sgd_update(p.y, d.y, coef, o.lr, o.wd);
      p.z = sgd_update(p.z, d.z, coef, o.lr, o.wd); p.w = sgd_update(p.w, d.w, coef, o.lr, o.wd);
    } else {
      p.x = adam_update(p.x, d.x, m.x, v.x, coef, o); p.y = adam_update(p.y, d.y, m.y, v.y, coef, o);
      p.z = adam_update(p.z, d.z, m.z, v.z, coef, o); p.w = adam_update(p.w, d.w, m.w, v.w, coef, o);
      *reinterpret_cast<float4 *>(o.m + i) = m;
      *reinterpret_cast<float4 *>(o.v + i) = v;
    }
    *reinterpret_cast<float4 *>(w + i) = p;
    if (o.zero_g) *reinterpret_cast<float4 *>(o.g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    return p;
  }
};
template <int KIND>
__device__ __forceinline__ float4 update4(float *__restrict__ w, int64_t i, const OptArgs &o, float coef) {
  Quad<KIND> q;
  q.load(w, i, o);
  return q.apply(w, i, o, coef);
}

struct PlanA {
  const ptamd_wprep_seg *segs;
  const int2 *blocks;        // (segment, block inside it); segment < 0: plain range -(segment + 1) of `plain`
  const int64_t *plain;      // [nplain][2]: first element, number of elements (multiples of 4; the tail of the buffer excepted)
  uint32_t *scales;          // row / column scales (the caller's `ints`)
  uint32_t *colmax;          // [2][ncolmax] column maxima (bit patterns), this call's copy selected below
  double *colsq;             // column sum-of-squares partials (fp64)
  float *stats;              // [2][nstats][4]
  int with_planes;
};

struct RowAcc {
  float4 cm[MAXV];      // column maxima of this wavefront's rows
  double cs[MAXV][4];   // column sums of squares (fp64: see below)
  float bn, bm;         // largest row norm / largest |w| of the rows that enter the statistics
};
// The RW rows of one wavefront.  NV float4 per lane and row (cols <= 256 NV); PIPE: the loads of row r + 1 are issued in front
// of the arithmetic and the two wavefront reductions of row r (short rows: two 16-byte requests per lane and stream would
// otherwise be all that is in flight - measured 3.9 TB/s for the SGD variant against 7 TB/s of the plain kernel).
template <int KIND, int NV, bool PIPE>
__device__ __forceinline__ void rows_loop(float *__restrict__ w, const PlanA &pl, const OptArgs &o, const ptamd_wprep_seg &sg, int row_first,
                                          int lane, float coef, bool want_cols, bool want_sq, char *planes, RowAcc &acc) {
  const int cols = sg.cols;
  Quad<KIND> cur[NV], nxt[NV];
  auto load = [&](Quad<KIND> (&q)[NV], int r) __attribute__((always_inline)) {
    const int64_t row0 = sg.offset + (int64_t)r * sg.ld;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < cols) q[i].load(w, row0 + c, o);
    }
  };
  if (row_first < sg.rows) load(cur, row_first);
#pragma unroll 1
  for (int rr = 0; rr < RW; ++rr) {
    const int r = row_first + rr;
    if (r >= sg.rows) break;  // (wavefront-uniform)
    if (PIPE && rr + 1 < RW && r + 1 < sg.rows) load(nxt, r + 1);
    const int64_t row0 = sg.offset + (int64_t)r * sg.ld;
    float4 v[NV];
    float m = 0.f, sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < cols) {
        v[i] = cur[i].apply(w, row0 + c, o, coef);
        // (row maximum and row sum of squares in the order of wscale_kernel: the statistics - and with them every bound - are
        // bit for bit what ptamd_weight_scales gives)
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[i].x), fabsf(v[i].y))), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
        sq = fmaf(v[i].x, v[i].x, fmaf(v[i].y, v[i].y, fmaf(v[i].z, v[i].z, fmaf(v[i].w, v[i].w, sq))));
        if (want_cols) {
          acc.cm[i].x = fmaxf(acc.cm[i].x, fabsf(v[i].x)); acc.cm[i].y = fmaxf(acc.cm[i].y, fabsf(v[i].y));
          acc.cm[i].z = fmaxf(acc.cm[i].z, fabsf(v[i].z)); acc.cm[i].w = fmaxf(acc.cm[i].w, fabsf(v[i].w));
        }
        if (want_sq) {
          acc.cs[i][0] = fma((double)v[i].x, (double)v[i].x, acc.cs[i][0]); acc.cs[i][1] = fma((double)v[i].y, (double)v[i].y, acc.cs[i][1]);
          acc.cs[i][2] = fma((double)v[i].z, (double)v[i].z, acc.cs[i][2]); acc.cs[i][3] = fma((double)v[i].w, (double)v[i].w, acc.cs[i][3]);
        }
      } else {
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    m = wave_max(m);
    sq = wave_sum(sq);
    const uint32_t sbits = pt_row_scale_bits(__float_as_uint(m));
    if (lane == 0 && sg.row_scale_index >= 0) pl.scales[sg.row_scale_index + r] = sbits;
    // a column panel of a wider matrix: the row's maximum meets those of the other panels in the pool of maxima (kernel B turns
    // it into the row scale); its statistics are the maximum only (stats_row0 < 0: a panel's norm is not the row's)
    if (lane == 0 && sg.rowmax_index >= 0) atomicMax(pl.colmax + sg.rowmax_index + r, __float_as_uint(m));
    if (sg.stats_row0 < 0) {
      acc.bm = fmaxf(acc.bm, m);
    } else if (r >= sg.stats_row0) {
      acc.bn = fmaxf(acc.bn, sqrtf(sq));
      acc.bm = fmaxf(acc.bm, m);
    }
    if (planes) {  // (rows and cols are multiples of 32 here: the builder of the plan checks)
      const float s = __uint_as_float(sbits);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) pthp::store4_split(planes, cols >> 4, r, c, v[i], s);
      }
    }
    if (PIPE) {
#pragma unroll
      for (int i = 0; i < NV; ++i) cur[i] = nxt[i];
    } else if (rr + 1 < RW && r + 1 < sg.rows) {
      load(cur, r + 1);
    }
  }
}

template <int KIND>
__global__ __launch_bounds__(64 * WAVES) void wprep_rows_kernel(float *__restrict__ w, const PlanA pl, const OptArgs o) {
  const int2 blk = pl.blocks[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float coef = KIND == 0 ? 1.f : clip_coef(o.sqnorm, o.max_norm);
  if (blk.x < 0) {  // ---- plain parameters: the optimizer update and nothing else
    if (KIND == 0) return;
    const int64_t first = pl.plain[2 * (-(blk.x + 1))], n = pl.plain[2 * (-(blk.x + 1)) + 1];
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blk.y * PLAIN_F4 + tid; i < min(n4, (int64_t)(blk.y + 1) * PLAIN_F4); i += 64 * WAVES)
      update4<KIND>(w, first + 4 * i, o, coef);
    if ((n & 3) && blk.y == 0 && tid < (int)(n & 3)) {  // (only the very end of a buffer whose length is not a multiple of 4)
      const int64_t k = first + (n4 << 2) + tid;
      if (KIND == 1) w[k] = sgd_update(w[k], o.g[k], coef, o.lr, o.wd);
      else w[k] = adam_update(w[k], o.g[k], o.m[k], o.v[k], coef, o);
      if (o.zero_g) o.g[k] = 0.f;
    }
    return;
  }
  const ptamd_wprep_seg sg = pl.segs[blk.x];
  const int cols = sg.cols, nv = (cols + 255) >> 8;
  __shared__ __attribute__((aligned(16))) float s_col[WAVES][2 * 64 * 4 * MAXV];   // column maxima (floats), then column sums of squares (doubles), per wavefront
  __shared__ float s_nrm[WAVES], s_amx[WAVES];
  const bool want_cols = sg.col_scale_index >= 0, want_sq = sg.colsq_index >= 0;
  // (column sums of squares in fp64: the largest column norm is then the same fp32 number whatever the order of the sum -
  // here per wavefront and 32-row block, in wscale_kernel per 64-row stride - and with it every bound derived from it)
  RowAcc acc;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    acc.cm[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    acc.cs[i][0] = acc.cs[i][1] = acc.cs[i][2] = acc.cs[i][3] = 0.0;
  }
  acc.bn = acc.bm = 0.f;
  char *planes = pl.with_planes ? reinterpret_cast<char *>(sg.row_planes) : nullptr;
  const int row_first = blk.y * RB + wave * RW;
  rows_loop<KIND, MAXV, true>(w, pl, o, sg, row_first, lane, coef, want_cols, want_sq, planes, acc);
  const float bn = acc.bn, bm = acc.bm;
  float4 (&cm)[MAXV] = acc.cm;
  double (&cs)[MAXV][4] = acc.cs;
  if (sg.stats_index >= 0) {
    if (lane == 0) {
      s_nrm[wave] = bn;
      s_amx[wave] = bm;
    }
  }
  if (want_cols) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (i < nv) *reinterpret_cast<float4 *>(&s_col[wave][(i * 64 + lane) * 4]) = cm[i];
  }
  __syncthreads();
  if (sg.stats_index >= 0 && tid == 0) {
    float n = 0.f, a = 0.f;
#pragma unroll
    for (int k = 0; k < WAVES; ++k) {
      n = fmaxf(n, s_nrm[k]);
      a = fmaxf(a, s_amx[k]);
    }
    uint32_t *st = reinterpret_cast<uint32_t *>(pl.stats) + 4 * sg.stats_index;
    atomicMax(st + 0, __float_as_uint(n));   // (non-negative floats order like their bit patterns)
    atomicMax(st + 2, __float_as_uint(a));
  }
  if (want_cols) {
    for (int c = tid; c < cols; c += 64 * WAVES) {
      const float mx = fmaxf(fmaxf(s_col[0][c], s_col[1][c]), fmaxf(s_col[2][c], s_col[3][c]));
      atomicMax(pl.colmax + sg.colmax_index + c, __float_as_uint(mx));
    }
  }
  if (want_sq) {
    // one fp64 partial per (32-row block, column): the four wavefronts meet in LDS, summed in wavefront order; kernel B adds the
    // blocks in block order
    double (*s_d)[64 * 4 * MAXV] = reinterpret_cast<double (*)[64 * 4 * MAXV]>(&s_col[0][0]);
    double *dst = pl.colsq + sg.colsq_index + (int64_t)blk.y * cols;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (i < nv && c < cols) {
        s_d[wave][c] = cs[i][0]; s_d[wave][c + 1] = cs[i][1]; s_d[wave][c + 2] = cs[i][2]; s_d[wave][c + 3] = cs[i][3];
      }
    }
    __syncthreads();
    for (int c = tid; c < cols; c += 64 * WAVES) dst[c] = ((s_d[0][c] + s_d[1][c]) + s_d[2][c]) + s_d[3][c];
  }
}

struct PlanB {
  const ptamd_wprep_seg *segs;
  const int4 *blocks;        // (type, index, local block, -): 0 column scales of segment `index`, 1 W^T planes of segment `index`,
                             // 2 bounds group `index`, 3 row scales of the panelled matrix `index`
  const ptamd_wprep_bound *bounds;
  const int4 *groups;        // per bounds group: (first bound job, number of jobs, first colnorm segment entry, number of entries)
  const int *colnorm_segs;   // segments whose largest column norm goes to their statistics record [1]
  uint32_t *scales;
  float *values;             // out_value targets of the bound jobs
  const uint32_t *colmax;    // this call's copy
  uint32_t *colmax_next;     // the other copy: zeroed here for the next call
  const double *colsq;
  float *stats;              // this call's copy (column norms are added here)
  float *stats_next;
  int nstats, with_planes;
};

__global__ __launch_bounds__(256) void wprep_cols_kernel(const float *__restrict__ w, const PlanB pl) {
  const int4 blk = pl.blocks[blockIdx.x];
  const int tid = threadIdx.x;
  if (blk.x == 0) {  // ---- column scales of one matrix (and the reset of the other copy of its maxima)
    const ptamd_wprep_seg sg = pl.segs[blk.y];
    for (int c = blk.z * 2048 + tid; c < min(sg.cols, (blk.z + 1) * 2048); c += 256) {
      pl.scales[sg.col_scale_index + c] = pt_row_scale_bits(pl.colmax[sg.colmax_index + c]);
      pl.colmax_next[sg.colmax_index + c] = 0u;
    }
    return;
  }
  if (blk.x == 3) {  // ---- row scales of a matrix that went through kernel A as column panels (and the reset of the other copy)
    const ptamd_wprep_seg sg = pl.segs[blk.y];
    const int r = blk.z * 256 + tid;
    if (r < sg.rows) {
      pl.scales[sg.row_scale_index + r] = pt_row_scale_bits(pl.colmax[sg.rowmax_index + r]);
      pl.colmax_next[sg.rowmax_index + r] = 0u;
    }
    return;
  }
  if (blk.x == 1) {  // ---- planes of the TRANSPOSE (operand rows = columns of the matrix), scaled by the column scales
    if (!pl.with_planes) return;
    const ptamd_wprep_seg sg = pl.segs[blk.y];
    const int rows = sg.cols, K = sg.rows, kbv = pthp::kb16(K), ld = sg.ld;     // operand [rows = cols of W][K = rows of W]
    const int64_t nchunks = (int64_t)(pthp::round_up(rows, 32) / 32) * kbv * 64;
    const int64_t id = (int64_t)blk.z * 256 + tid;
    if (id >= nchunks) return;
    const int c = (int)(id & 63);
    const int64_t b = id >> 6;
    const int kb = (int)(b % kbv), rb = (int)(b / kbv);
    int r, h;
    pthp::chunk_coords(c, r, h);
    const int row = rb * 32 + r, k0 = kb * 16 + h * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    float sc = 1.f;
    if (row < rows) {
      sc = __uint_as_float(pt_row_scale_bits(pl.colmax[sg.colmax_index + row]));
      const float *x = w + sg.offset;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (k0 + e < K) v[e] = x[(size_t)(k0 + e) * ld + row];
    }
    uint4 hi, lo;
    ptsplit::split_pair_f16(v[0], v[1], sc, sc, hi.x, lo.x);
    ptsplit::split_pair_f16(v[2], v[3], sc, sc, hi.y, lo.y);
    ptsplit::split_pair_f16(v[4], v[5], sc, sc, hi.z, lo.z);
    ptsplit::split_pair_f16(v[6], v[7], sc, sc, hi.w, lo.w);
    char *dst = reinterpret_cast<char *>(sg.col_planes) + pthp::block_offset(rb, kb, 0, kbv) + c * 16;
    *reinterpret_cast<uint4 *>(dst) = hi;
    *reinterpret_cast<uint4 *>(dst + pthp::BLK_BYTES) = lo;
    return;
  }
  // ---- one bounds group (an encoder layer): largest column norm of its flagged matrices, then its bound jobs
  const int4 grp = pl.groups[blk.y];
  __shared__ float s_red[4];
  for (int e = 0; e < grp.w; ++e) {
    const ptamd_wprep_seg sg = pl.segs[pl.colnorm_segs[grp.z + e]];
    const int nb = (sg.rows + RB - 1) / RB;
    float nmax = 0.f;
    for (int c = tid; c < sg.cols; c += 256) {
      // (fixed order: four interleaved partial sums over the 32-row blocks - eight independent loads in flight per round instead of
      // one load per addition of a single chain, which made this the slowest part of the launch)
      const double *q = pl.colsq + sg.colsq_index + c;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int b = 0;
      for (; b + 8 <= nb; b += 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = q[(int64_t)(b + u) * sg.cols];
        s0 += t[0]; s1 += t[1]; s2 += t[2]; s3 += t[3];
        s0 += t[4]; s1 += t[5]; s2 += t[6]; s3 += t[7];
      }
      for (; b < nb; ++b) s0 += q[(int64_t)b * sg.cols];
      nmax = fmaxf(nmax, (float)sqrt((s0 + s1) + (s2 + s3)));
    }
    nmax = wave_max(nmax);
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = nmax;
    __syncthreads();
    // (several entries may share a record - the panels of one matrix: the largest of them, one after the other)
    if (tid == 0) pl.stats[4 * sg.stats_index + 1] = fmaxf(pl.stats[4 * sg.stats_index + 1], fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3])));
  }
  __syncthreads();  // (the store above is read below by threads of this workgroup only)
  __threadfence_block();
  if (tid < grp.y) {
    const ptamd_wprep_bound b = pl.bounds[grp.x + tid];
    float x = 1.f;  // bound of the L2 norm of the input row (bound_kernel of scales.hip, same expression)
    if (b.ln_gamma_stats >= 0) x = pl.stats[4 * b.ln_gamma_stats + 2] * b.sqrt_d + (b.ln_beta_stats >= 0 ? pl.stats[4 * b.ln_beta_stats] : 0.f);
    float v = b.w_stats >= 0 ? x * pl.stats[4 * b.w_stats + b.w_stat_index] : x;   // (no weight: the bound of the input row itself)
    if (b.bias_stats >= 0) v += pl.stats[4 * b.bias_stats + 2];
    v *= b.post_scale;
    if (b.out_scale >= 0) {
      const uint32_t s = pt_row_scale_bits(__float_as_uint(v));
      pl.scales[b.out_scale] = pl.scales[b.out_scale + 1] = pl.scales[b.out_scale + 2] = pl.scales[b.out_scale + 3] = s;
    }
    if (b.out_value >= 0) pl.values[b.out_value] = v;
  }
  if (blk.y == 0)   // the other copy of the statistics, for the next call
    for (int i = tid; i < 4 * pl.nstats; i += 256) pl.stats_next[i] = 0.f;
}

int run(const ptamd_wprep_plan *p, float *w, int parity, const OptArgs &o, hipStream_t st) {
  if (!p || !w || !p->segs || !p->blocks_a || !p->blocks_b || !p->scales || !p->colmax || !p->stats) return PTAMD_ERR_BAD_SHAPE;
  if (p->nblocks_a <= 0 || p->nblocks_b <= 0 || p->nstats <= 0 || (parity != 0 && parity != 1)) return PTAMD_ERR_BAD_SHAPE;
  if (!pt_aligned16(w) || (o.kind && !pt_aligned16(o.g)) || (o.kind == 2 && (!pt_aligned16(o.m) || !pt_aligned16(o.v)))) return PTAMD_ERR_ALIGN;
  PlanA a;
  a.segs = p->segs;
  a.blocks = reinterpret_cast<const int2 *>(p->blocks_a);
  a.plain = p->plain;
  a.scales = p->scales;
  a.colmax = p->colmax + (size_t)parity * p->ncolmax;
  a.colsq = reinterpret_cast<double *>(p->colsq);
  a.stats = p->stats + (size_t)parity * 4 * p->nstats;
  a.with_planes = p->with_planes;
  // (without an update the plain workgroups - listed behind the matrix ones - are not launched)
  const int grid_a = o.kind ? p->nblocks_a : p->nblocks_a_matrices;
  if (grid_a <= 0) return PTAMD_ERR_BAD_SHAPE;
  if (o.kind == 0) hipLaunchKernelGGL(wprep_rows_kernel<0>, dim3(grid_a), dim3(64 * WAVES), 0, st, w, a, o);
  else if (o.kind == 1) hipLaunchKernelGGL(wprep_rows_kernel<1>, dim3(grid_a), dim3(64 * WAVES), 0, st, w, a, o);
  else hipLaunchKernelGGL(wprep_rows_kernel<2>, dim3(grid_a), dim3(64 * WAVES), 0, st, w, a, o);
  int rc = pt_check_launch();
  if (rc) return rc;
  PlanB b;
  b.segs = p->segs;
  b.blocks = reinterpret_cast<const int4 *>(p->blocks_b);
  b.bounds = p->bounds;
  b.groups = reinterpret_cast<const int4 *>(p->groups);
  b.colnorm_segs = p->colnorm_segs;
  b.scales = p->scales;
  b.values = p->values;
  b.colmax = a.colmax;
  b.colmax_next = p->colmax + (size_t)(1 - parity) * p->ncolmax;
  b.colsq = reinterpret_cast<const double *>(p->colsq);
  b.stats = a.stats;
  b.stats_next = p->stats + (size_t)(1 - parity) * 4 * p->nstats;
  b.nstats = p->nstats;
  b.with_planes = p->with_planes;
  hipLaunchKernelGGL(wprep_cols_kernel, dim3(p->nblocks_b), dim3(256), 0, st, w, b);
  return pt_check_launch();
}

}  // namespace

extern "C" {

int ptamd_wprep_rows_per_block(void) { return RB; }
int ptamd_wprep_plain_floats_per_block(void) { return 4 * PLAIN_F4; }

int ptamd_weights_prep(const ptamd_wprep_plan *plan, const float *w, int parity, void *stream) {
  OptArgs o = {};
  o.kind = 0;
  return run(plan, const_cast<float *>(w), parity, o, (hipStream_t)stream);
}

int ptamd_sgd_step_prep(const ptamd_wprep_plan *plan, int parity, float *w, float *g, int64_t n, const float *sqnorm,
                        float max_norm, float lr, float weight_decay, int zero_grad, void *stream) {
  if (!plan || n != plan->numel || !g) return PTAMD_ERR_BAD_SHAPE;
  OptArgs o = {};
  o.kind = 1; o.g = g; o.zero_g = zero_grad != 0; o.sqnorm = sqnorm; o.max_norm = max_norm; o.lr = lr; o.wd = weight_decay;
  return run(plan, w, parity, o, (hipStream_t)stream);
}

int ptamd_adam_step_prep(const ptamd_wprep_plan *plan, int parity, float *w, float *g, float *m, float *v, int64_t n,
                         const float *sqnorm, float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                         int step, int zero_grad, void *stream) {
  if (!plan || n != plan->numel || !g || !m || !v || step <= 0) return PTAMD_ERR_BAD_SHAPE;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  OptArgs o = {};
  o.kind = 2; o.g = g; o.zero_g = zero_grad != 0; o.m = m; o.v = v; o.sqnorm = sqnorm; o.max_norm = max_norm; o.wd = weight_decay;
  o.beta1 = beta1; o.beta2 = beta2; o.eps = eps; o.step_size = (float)(lr / bc1); o.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  return run(plan, w, parity, o, (hipStream_t)stream);
}

}  // extern "C"
