// Batched dRMSD loss, forward + analytic backward, for gfx950.
//
// Replaces, for a whole batch on the device, the per-protein CPU chain of
//   drmsd_work (loss part)      /root/reference/protein_transformer/losses.py:63-92
//   pairwise_internal_dist      .../losses.py:233-253   (n x n matrices are never materialised here)
//   drmsd                       .../losses.py:256-278
//   get_backbone_from_full_coords  .../protein/structure_utils.py:19-32
// and the autograd backward of `l_normed = drmsd / n` (losses.py:80,91-92):
//   d(l)/dx_i = 1/(n P D) * sum_{j != i} (d_ij - t_ij)/d_ij (x_i - x_j),   P = n(n-1)/2   
//
// Three launches per batch:
//   compact   NaN-mask compaction of (pred, true) atoms per protein; backbone atoms (slots 0..2) are packed
//             FIRST so the backbone-only dRMSD falls out of the same sweep (dRMSD is permutation invariant).
//   pairs     grid (row block, protein): each lane owns one atom i, column tiles of 256 atoms are staged in
//             LDS and read as broadcasts; per pair 2 transcendentals (one v_rsq_f32 each for the predicted and the true
//             distance; the predicted one doubles as 1/d for the gradient).  ALU/transcendental bound, O(n) bytes.
//   finalize  fixed-order fp64 reduction of the block partials, loss statistics, gradient scale and scatter
//             back to the [L*14,3] slot layout.
#include "common.h"

namespace {

constexpr int CB = 256;  // threads per block of the compaction / finalize kernels
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef PT_DRMSD_PB
#define PT_DRMSD_PB 128
#endif
#ifndef PT_DRMSD_UNROLL
#define PT_DRMSD_UNROLL 4
#endif
constexpr int PB = PT_DRMSD_PB;  // rows (= threads) per block of the pair sweep: ~34 blocks per protein keep 256 CUs balanced

struct Counts {
  int n, n_bb, len, pad;
};

__device__ int protein_len_block(const int64_t *seq, int L, int *s_tmp) {
  int cnt = 0;
  for (int i = threadIdx.x; i < L; i += blockDim.x) cnt += (seq[i] != PTAMD_PAD_ID);
  cnt = (int)wave_sum((float)cnt);
  if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = cnt;
  __syncthreads();
  int tot = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += s_tmp[w];
  __syncthreads();
  return tot;
}

__global__ __launch_bounds__(CB) void drmsd_compact_kernel(const float *__restrict__ pred,
                                                           const float *__restrict__ truth,
                                                           const int64_t *__restrict__ seq, int L,
                                                           float4 *__restrict__ pred4, float4 *__restrict__ true4,
                                                           int *__restrict__ idx, Counts *__restrict__ counts) {
  __shared__ int s_bb[CB], s_ot[CB], s_tmp[CB / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const size_t nmax = (size_t)L * 14;
  pred += (size_t)b * nmax * 3;
  truth += (size_t)b * nmax * 3;
  pred4 += (size_t)b * nmax;
  true4 += (size_t)b * nmax;
  idx += (size_t)b * nmax;
  const int len = protein_len_block(seq + (size_t)b * L, L, s_tmp);
  const int nslot = len * 14;
  const int per = (nslot + CB - 1) / CB;
  const int s0 = tid * per, s1 = min(s0 + per, nslot);
  int nbb = 0, not_ = 0;
  for (int s = s0; s < s1; ++s) {
    float tx = truth[s * 3], ty = truth[s * 3 + 1], tz = truth[s * 3 + 2];
    bool ok = !(isnan(tx) || isnan(ty) || isnan(tz));
    bool bb = (s % 14) < 3;
    nbb += ok && bb;
    not_ += ok && !bb;
  }
  s_bb[tid] = nbb;
  s_ot[tid] = not_;
  __syncthreads();
  // exclusive scan over 256 entries; small enough that every thread just sums its prefix
  int off_bb = 0, off_ot = 0, tot_bb = 0, tot_ot = 0;
  for (int t = 0; t < CB; ++t) {
    int vb = s_bb[t], vo = s_ot[t];
    if (t < tid) {
      off_bb += vb;
      off_ot += vo;
    }
    tot_bb += vb;
    tot_ot += vo;
  }
  int pb = off_bb, po = tot_bb + off_ot;
  for (int s = s0; s < s1; ++s) {
    float tx = truth[s * 3], ty = truth[s * 3 + 1], tz = truth[s * 3 + 2];
    bool ok = !(isnan(tx) || isnan(ty) || isnan(tz));
    if (!ok) continue;
    bool bb = (s % 14) < 3;
    int pos = bb ? pb++ : po++;
    pred4[pos] = make_float4(pred[s * 3], pred[s * 3 + 1], pred[s * 3 + 2], 0.f);
    true4[pos] = make_float4(tx, ty, tz, 0.f);
    idx[pos] = s;
  }
  if (tid == 0) counts[b] = Counts{tot_bb + tot_ot, tot_bb, len, 0};
}

template <bool WITH_GRAD>
__global__ __launch_bounds__(PB) void drmsd_pairs_kernel(const float4 *__restrict__ pred4,
                                                         const float4 *__restrict__ true4,
                                                         const Counts *__restrict__ counts, int L, int row_blocks,
                                                         float4 *__restrict__ gcomp, double *__restrict__ partials) {
  // column tile in LDS, predicted and true coordinate side by side: (px, tx, py, ty) and (pz, tz), so that the two
  // distance computations of a pair run as ONE stream of packed f32 instructions (v_pk_add / v_pk_fma_f32: two lanes'
  // worth of arithmetic per issue slot - the pair loop is VALU-bound, not memory-bound)
  __shared__ float4 s_xy[PB];
  __shared__ float2 s_z[PB];
  __shared__ double s_red[2 * (PB / 64)];
  const int b = blockIdx.y, tid = threadIdx.x;
  const Counts cn = counts[b];
  const int n = cn.n, nbb = cn.n_bb;
  double *part = partials + ((size_t)b * row_blocks + blockIdx.x) * 2;
  const int row0 = blockIdx.x * PB;
  if (row0 >= n) {  // block-uniform
    if (tid == 0) part[0] = part[1] = 0.0;
    return;
  }
  const size_t nmax = (size_t)L * 14;
  pred4 += (size_t)b * nmax;
  true4 += (size_t)b * nmax;
  const int i = row0 + tid;
  const bool live = i < n;
  const float4 pi = live ? pred4[i] : make_float4(0, 0, 0, 0);
  const float4 ti = live ? true4[i] : make_float4(0, 0, 0, 0);
  const f32x2 ix = {pi.x, ti.x}, iy = {pi.y, ti.y}, iz = {pi.z, ti.z};
  float accA = 0.f, accB = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;

  auto pair = [&](int j, float &acc) {
    const float4 a = s_xy[j];  // same address in every lane: LDS broadcast
    const float2 c = s_z[j];
    const f32x2 dx = ix - (f32x2){a.x, a.y}, dy = iy - (f32x2){a.z, a.w}, dz = iz - (f32x2){c.x, c.y};  // (pred, true)
    f32x2 q = dx * dx;
    q = __builtin_elementwise_fma(dy, dy, q);
    q = __builtin_elementwise_fma(dz, dz, q);
    const float d2 = fmaxf(q[0], 1e-30f), t2 = fmaxf(q[1], 1e-30f);  // clamp_min(1e-30) of losses.py:252
    const float inv = __builtin_amdgcn_rsqf(d2), invt = __builtin_amdgcn_rsqf(t2);
    f32x2 dt = (f32x2){d2, t2} * (f32x2){inv, invt};  // (d, tau)
    asm("" : "+v"(dt));  // keep the rounded products: no FMA contraction into e, so pred == true gives e == 0 exactly
    const float e = dt[0] - dt[1];
    acc = fmaf(e, e, acc);
    if (WITH_GRAD) {
      const float cf = e * inv;
      gx = fmaf(cf, dx[0], gx);
      gy = fmaf(cf, dy[0], gy);
      gz = fmaf(cf, dz[0], gz);
    }
  };

  for (int c0 = 0; c0 < n; c0 += PB) {
    __syncthreads();
    if (c0 + tid < n) {
      const float4 pj = pred4[c0 + tid], tj = true4[c0 + tid];
      s_xy[tid] = make_float4(pj.x, tj.x, pj.y, tj.y);
      s_z[tid] = make_float2(pj.z, tj.z);
    }
    __syncthreads();
    const int cnt = min(PB, n - c0);
    const int ja = max(0, min(cnt, nbb - c0));
    // four pairs per iteration, written out by hand (the optimizer declines to unroll this loop by itself): the loop
    // bookkeeping is shared and the next pairs' LDS reads are in flight under the current pair's arithmetic
    int j = 0;
    constexpr int U = PT_DRMSD_UNROLL;
    for (; j + U - 1 < ja; j += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) pair(j + u, accA);
    }
    for (; j < ja; ++j) pair(j, accA);
    for (; j + U - 1 < cnt; j += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) pair(j + u, accB);
    }
    for (; j < cnt; ++j) pair(j, accB);
  }
  if (WITH_GRAD && live) gcomp[(size_t)b * nmax + i] = make_float4(gx, gy, gz, 0.f);
  // every pair (i,j), i != j, was visited twice over the grid; j == i contributes exactly 0 to everything
  // (dx = 0 -> d = 1e-15, true distance 1e-15, e = 0).
  double all = live ? (double)accA + (double)accB : 0.0;
  double bbp = (live && i < nbb) ? (double)accA : 0.0;
  all = wave_sum_d(all);
  bbp = wave_sum_d(bbp);
  if ((tid & 63) == 0) {
    s_red[(tid >> 6) * 2] = all;
    s_red[(tid >> 6) * 2 + 1] = bbp;
  }
  __syncthreads();
  if (tid == 0) {
    double a = 0, c = 0;
    for (int w = 0; w < PB / 64; ++w) {
      a += s_red[w * 2];
      c += s_red[w * 2 + 1];
    }
    part[0] = a;
    part[1] = c;
  }
}

__global__ __launch_bounds__(CB) void drmsd_finalize_kernel(const Counts *__restrict__ counts,
                                                            const double *__restrict__ partials, int row_blocks,
                                                            const float4 *__restrict__ gcomp,
                                                            const int *__restrict__ idx, int L,
                                                            float *__restrict__ stats, float *__restrict__ dcrd) {
  __shared__ float s_scale;
  const int b = blockIdx.x, tid = threadIdx.x;
  const Counts cn = counts[b];
  if (tid == 0) {
    double all = 0, bbp = 0;
    for (int r = 0; r < row_blocks; ++r) {
      all += partials[((size_t)b * row_blocks + r) * 2];
      bbp += partials[((size_t)b * row_blocks + r) * 2 + 1];
    }
    const double n = cn.n, nb = cn.n_bb;
    const double P = n * (n - 1) * 0.5, Pb = nb * (nb - 1) * 0.5;
    // mse_loss over an empty pair set is NaN in the reference as well
    float D = (float)sqrt((all * 0.5) / P);
    float Db = (float)sqrt((bbp * 0.5) / Pb);
    float *st = stats + (size_t)b * 8;
    st[0] = D;
    st[1] = D / (float)cn.n;
    st[2] = Db;
    st[3] = Db / (float)cn.n_bb;
    st[4] = (float)cn.n;
    st[5] = (float)cn.n_bb;
    st[6] = 0.f;
    st[7] = 0.f;
    s_scale = (float)(1.0 / (n * P * (double)D));
  }
  if (dcrd == nullptr) return;
  __syncthreads();
  const float scale = s_scale;
  const size_t nmax = (size_t)L * 14;
  float *out = dcrd + (size_t)b * nmax * 3;
  for (size_t k = tid; k < nmax * 3; k += CB) out[k] = 0.f;
  __syncthreads();
  gcomp += (size_t)b * nmax;
  idx += (size_t)b * nmax;
  for (int j = tid; j < cn.n; j += CB) {
    const float4 g = gcomp[j];
    const int s = idx[j];
    out[s * 3 + 0] = scale * g.x;
    out[s * 3 + 1] = scale * g.y;
    out[s * 3 + 2] = scale * g.z;
  }
}

struct Layout {
  size_t pred4, true4, gcomp, idx, counts, partials, total;
  int row_blocks;
};
Layout layout(int B, int L) {
  Layout l;
  const size_t nmax = (size_t)L * 14, BN = (size_t)B * nmax;
  l.row_blocks = (int)((nmax + PB - 1) / PB);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  l.pred4 = take(BN * sizeof(float4));
  l.true4 = take(BN * sizeof(float4));
  l.gcomp = take(BN * sizeof(float4));
  l.idx = take(BN * sizeof(int));
  l.counts = take((size_t)B * sizeof(Counts));
  l.partials = take((size_t)B * l.row_blocks * 2 * sizeof(double));
  l.total = off;
  return l;
}

}  // namespace

extern "C" {

size_t ptamd_drmsd_workspace_bytes(int B, int L) {
  if (B <= 0 || L <= 0) return 0;
  return layout(B, L).total;
}

int ptamd_drmsd_fwd_bwd(const float *pred_crd, const float *true_crd, const int64_t *seq, int B, int L, float *stats,
                        float *dcrd, void *workspace, size_t workspace_bytes, void *stream) {
  if (B <= 0 || L <= 0) return PTAMD_ERR_BAD_SHAPE;
  const Layout l = layout(B, L);
  if (!workspace || workspace_bytes < l.total) return PTAMD_ERR_WORKSPACE;
  if (!pt_aligned16(workspace)) return PTAMD_ERR_ALIGN;
  char *ws = static_cast<char *>(workspace);
  float4 *pred4 = reinterpret_cast<float4 *>(ws + l.pred4), *true4 = reinterpret_cast<float4 *>(ws + l.true4),
         *gcomp = reinterpret_cast<float4 *>(ws + l.gcomp);
  int *idx = reinterpret_cast<int *>(ws + l.idx);
  Counts *counts = reinterpret_cast<Counts *>(ws + l.counts);
  double *partials = reinterpret_cast<double *>(ws + l.partials);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(drmsd_compact_kernel, dim3(B), dim3(CB), 0, st, pred_crd, true_crd, seq, L, pred4, true4, idx,
                     counts);
  int rc = pt_check_launch();
  if (rc) return rc;
  if (dcrd)
    hipLaunchKernelGGL(drmsd_pairs_kernel<true>, dim3(l.row_blocks, B), dim3(PB), 0, st, pred4, true4, counts, L,
                       l.row_blocks, gcomp, partials);
  else
    hipLaunchKernelGGL(drmsd_pairs_kernel<false>, dim3(l.row_blocks, B), dim3(PB), 0, st, pred4, true4, counts, L,
                       l.row_blocks, gcomp, partials);
  rc = pt_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(drmsd_finalize_kernel, dim3(B), dim3(CB), 0, st, counts, partials, l.row_blocks, gcomp, idx, L,
                     stats, dcrd);
  return pt_check_launch();
}

}  // extern "C"
