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
//   pairs     the UPPER TRIANGLE of the pair matrix, every unordered pair once (drmsd_tri_kernel below): each lane owns one
//             row atom i, the column atom of a step is wavefront-uniform and arrives by SCALAR loads (its coordinates are
//             SGPR operands of the packed subtractions - no LDS, no vector registers); per pair 2 transcendentals
//             (one v_rsq_f32 each for the predicted and the true distance; the predicted one doubles as 1/d for the
//             gradient); the column atoms' share of the gradient comes from a transposed re-read of the coefficient
//             tile through LDS.  ALU/transcendental bound, O(n) bytes + O(n^2 / 256) partial-sum bytes.
//   finalize  fixed-order fp64 reduction of the block partials, loss statistics, gradient assembly (row + column
//             partials, fixed order), scale and scatter back to the [L*14,3] slot layout.
#include <type_traits>
#include <utility>

#include <stdlib.h>

#include "common.h"

namespace {

constexpr int CB = 256;  // threads per block of the compaction / finalize kernels
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef PT_DRMSD_UNROLL
#define PT_DRMSD_UNROLL 8   // column atoms per batch = independent chains in flight per wavefront (4: the same time)
#endif

struct Counts {
  int n, n_bb, len, pad;
};
// a compacted atom as the pair sweep wants a COLUMN atom: (predicted, true) interleaved per axis - the pairs are SGPR operands
// of v_pk_add_f32 - one s_load_dwordx4 + one s_load_dwordx2 per atom
struct __attribute__((aligned(32))) Col8 {
  float px, tx, py, ty, pz, tz, r0, r1;
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

// Compaction of one protein's atom slots (14 per residue, NaN = absent) into dense arrays, backbone atoms first: one
// workgroup per protein, each of its 16 wavefronts owns a contiguous share of the slots and walks it 64 slots at a time
// (coalesced loads, position of a present atom = running count + rank among the lanes before it: v_mbcnt of the ballot) -
// no workgroup barrier inside the loops, so the loads of consecutive rows overlap (the old per-thread runs of 28 slots
// were 2 x 28 dependent strided loads: 46 us whatever the batch).  Order inside the two classes: slot order.
constexpr int COMPACT_THREADS = 1024;  // 16 wavefronts: 7 rows of 64 slots each at L = 512
__global__ __launch_bounds__(COMPACT_THREADS) void drmsd_compact_kernel(const float *__restrict__ pred,
                                                           const float *__restrict__ truth,
                                                           const int64_t *__restrict__ seq, int L,
                                                           float4 *__restrict__ pred4, Col8 *__restrict__ col8,
                                                           int *__restrict__ idx, Counts *__restrict__ counts) {
  constexpr int NWAVE = COMPACT_THREADS / 64;
  __shared__ int s_bb[NWAVE], s_ot[NWAVE], s_tmp[NWAVE];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const size_t nmax = (size_t)L * 14;
  pred += (size_t)b * nmax * 3;
  truth += (size_t)b * nmax * 3;
  pred4 += (size_t)b * nmax;
  col8 += (size_t)b * nmax;
  idx += (size_t)b * nmax;
  const int len = protein_len_block(seq + (size_t)b * L, L, s_tmp);
  const int nslot = len * 14;
  const int per = ((nslot + NWAVE - 1) / NWAVE + 63) / 64 * 64;   // slots of a wavefront: whole rows of 64
  const int s0 = min(w * per, nslot), s1 = min(s0 + per, nslot);
  auto present = [&](int s, float &tx, float &ty, float &tz) __attribute__((always_inline)) {
    const int sc = min(s, max(nslot - 1, 0));
    tx = truth[sc * 3]; ty = truth[sc * 3 + 1]; tz = truth[sc * 3 + 2];
    return s < s1 && !(isnan(tx) || isnan(ty) || isnan(tz));
  };
  int nbb = 0, not_ = 0;   // (wavefront-uniform)
  for (int r = s0; r < s1; r += 64) {
    float tx, ty, tz;
    const int s = r + lane;
    const bool ok = present(s, tx, ty, tz), bb = (s % 14) < 3;
    nbb += __popcll(__ballot(ok && bb));
    not_ += __popcll(__ballot(ok && !bb));
  }
  if (lane == 0) {
    s_bb[w] = nbb;
    s_ot[w] = not_;
  }
  __syncthreads();
  int pb = 0, po = 0, tot_bb = 0, tot_ot = 0;
#pragma unroll
  for (int t = 0; t < NWAVE; ++t) {
    if (t < w) {
      pb += s_bb[t];
      po += s_ot[t];
    }
    tot_bb += s_bb[t];
    tot_ot += s_ot[t];
  }
  po += tot_bb;
  for (int r = s0; r < s1; r += 64) {
    float tx, ty, tz;
    const int s = r + lane;
    const bool ok = present(s, tx, ty, tz), bb = (s % 14) < 3;
    const unsigned long long mb = __ballot(ok && bb), mo = __ballot(ok && !bb);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (ok) {
      const int pos = bb ? pb + __popcll(mb & below) : po + __popcll(mo & below);
      const float px = pred[s * 3], py = pred[s * 3 + 1], pz = pred[s * 3 + 2];
      pred4[pos] = make_float4(px, py, pz, 0.f);
      col8[pos] = Col8{px, tx, py, ty, pz, tz, 0.f, 0.f};
      idx[pos] = s;
    }
    pb += __popcll(mb);
    po += __popcll(mo);
  }
  if (tid == 0) counts[b] = Counts{tot_bb + tot_ot, tot_bb, len, 0};
}

// ---- the pair sweep over the UPPER TRIANGLE: every unordered pair {i, j} is evaluated ONCE.
// Atoms are cut into tiles of 64 (one wavefront's lanes); a workgroup of 4 wavefronts owns a STRIP of 4 row tiles
// (256 atoms) and walks a CHUNK of up to 8 column tiles J >= its first tile (the chunks of a strip are separate work
// items: a strip's work shrinks with its index, and a CU holds three workgroups at a time - with 16-tile chunks the
// benchmark batch was 1824 items, 7 per CU in 2.4 rounds of 3, and the CUs that drew a third round set the time: 8-tile
// chunks 315 us against 352 for the whole loss, 4-tile chunks 320 - the partials of the finalize kernel double each time).  For a column tile J above a wavefront's row tile I (I < J) the wavefront
//   phase 1  walks 16 columns like the old two-sided kernel did (lane = row atom i, column coordinates broadcast from
//            LDS, one v_rsq_f32 each for the predicted and the true distance), adds e^2 to the loss, cf (x_i - x_j) to
//            its row gradient - and leaves the coefficient cf_ij = e / d in LDS, row-major with a 17-word row stride;
//   phase 2  re-reads that 64 x 16 coefficient tile TRANSPOSED (lane = column j = lane & 15 and a quarter lane >> 4 of the
//            rows; conflict-free both ways thanks to the odd stride), accumulates S_j = sum_i cf_ij and
//            V_j = sum_i cf_ij x_i over its 16 rows and folds the four quarters with two lane exchanges: 4 fma + one LDS
//            read per pair instead of the ~17 instructions and two transcendentals of a second visit.  The column atom's
//            gradient contribution is x_j S_j - V_j.  (A 64-column coefficient tile would be 16.6 KB per wavefront and
//            hold the kernel at 2 wavefronts per SIMD: measured 510 us, latency-bound, against 576 for the old kernel.)
// The four wavefronts' (S, V) of a column tile are summed in a fixed order through LDS and written to a per-(strip,
// column tile) slot; the finalize kernel adds, per atom and in a fixed order, the row partials of its strip's chunks and
// the column partials of all strips at or above it: no atomics anywhere, bit-reproducible.  The diagonal tile (I == J) is
// still swept from both sides inside the tile (its loss terms count half).
constexpr int TS = 64, STRIP_TILES = 4, RS = TS * STRIP_TILES;
// column tiles per work item: 8, or fewer (4) when the batch would otherwise leave most of the 5 x 256 workgroup slots
// empty (few or short proteins) - a function of (B, L) only, so a given batch is always cut the same way
#ifndef PT_DRMSD_MAX_CHUNK
#define PT_DRMSD_MAX_CHUNK 8
#endif
constexpr int MAX_CHUNK_TILES = PT_DRMSD_MAX_CHUNK, MIN_CHUNK_TILES = 4, TARGET_ITEMS = 2560;
constexpr int SUB = 16, CF_LD = SUB + 1;   // the coefficient tile is kept for 16 columns at a time: 4.3 KB per wavefront

struct TriLayout {  // per protein: strips x chunks work items, column tiles
  int strips, chunks, tiles, chunk_tiles;
};
__host__ __device__ inline TriLayout tri_layout(int nmax, int B) {
  TriLayout t;
  t.tiles = (nmax + TS - 1) / TS;
  t.strips = (t.tiles + STRIP_TILES - 1) / STRIP_TILES;
  t.chunk_tiles = MAX_CHUNK_TILES;
  for (;;) {
    t.chunks = (t.tiles + t.chunk_tiles - 1) / t.chunk_tiles;
    const long items = (long)t.strips * (t.chunks + 1) / 2 * B;   // (strip, chunk) pairs of the upper triangle, about
    if (items >= TARGET_ITEMS || t.chunk_tiles <= MIN_CHUNK_TILES) break;
    t.chunk_tiles >>= 1;
  }
  return t;
}

// acc += (value of `rowval` in lane r of this lane's row of 16 lanes) * c: the DPP broadcast rides on the multiply-add
template <int R>
__device__ __forceinline__ void fmac_row16(float &acc, float rowval, float c) {
  asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(rowval), "v"(c), "n"(R));
}
// one row r of phase 2: q += cf[r] * (1, x_r, y_r, z_r), the row atom's coordinates straight from the lane that owns it
template <int R>
__device__ __forceinline__ void colsum_row(float4 &q, const float *__restrict__ cf_q, float px, float py, float pz) {
  const float c = cf_q[R * CF_LD];
  q.x += c;
  fmac_row16<R>(q.y, px, c);
  fmac_row16<R>(q.z, py, c);
  fmac_row16<R>(q.w, pz, c);
}
template <int... R>
__device__ __forceinline__ void colsum_rows(float4 &q, const float *__restrict__ cf_q, float px, float py, float pz,
                                            std::integer_sequence<int, R...>) {
  (colsum_row<R>(q, cf_q, px, py, pz), ...);
}

template <bool WITH_GRAD>
__global__ __launch_bounds__(RS) void drmsd_tri_kernel(const Col8 *__restrict__ col8,
                                                       const Counts *__restrict__ counts, int L,
                                                       float4 *__restrict__ rowpart, float4 *__restrict__ colpart,
                                                       double *__restrict__ partials, int strip0, int spp) {
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  float *const s_cf = s_dyn;                                                  // [4][64][17] coefficient tiles
  float4 *const s_cs = reinterpret_cast<float4 *>(s_dyn + STRIP_TILES * TS * CF_LD);   // [2][4][64] (S, Vx, Vy, Vz) per wavefront, two column tiles
  __shared__ double s_red[2 * STRIP_TILES];
  __shared__ Col8 s_col[2][TS];   // the column tile (all four wavefronts walk the same one), two buffers
  const size_t nmax = (size_t)L * 14;
  const TriLayout tl = tri_layout((int)nmax, (int)gridDim.y);
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // (told to the compiler: what depends on w only stays scalar)
  // this launch covers the strips strip0 .. strip0 + spp - 1 (all of them, or one PASS of a long batch: layout()); the row /
  // column partials are indexed by the strip's position in the pass, the loss partials by the strip itself
  const int sl = blockIdx.x / tl.chunks, strip = strip0 + sl, chunk = blockIdx.x % tl.chunks;
  const Counts cn = counts[b];
  const int n = cn.n, nbb = cn.n_bb, nT = (n + TS - 1) / TS;
  double *part = partials + ((size_t)b * tl.strips * tl.chunks + (size_t)strip * tl.chunks + chunk) * 2;
  const int J0 = max(STRIP_TILES * strip, tl.chunk_tiles * chunk), J1 = min(nT, tl.chunk_tiles * (chunk + 1));
  if (STRIP_TILES * strip >= nT || J0 >= J1) {  // block-uniform: nothing to do (the finalize kernel skips these items too)
    if (tid == 0) part[0] = part[1] = 0.0;
    return;
  }
  col8 += (size_t)b * nmax;
  const int I = STRIP_TILES * strip + w, i = I * TS + lane;
  const bool live = i < n;
  // A dead row (behind the protein's last atom) sits at (2^60, 2^60, 2^60) in BOTH structures: every difference to a real
  // atom rounds to 2^60 in both, d == tau exactly, e = cf = 0 - no select per pair, nothing non-finite anywhere.
  constexpr float FAR = 1152921504606846976.f;
  Col8 me = Col8{FAR, FAR, FAR, FAR, FAR, FAR, 0, 0};
  if (live) me = col8[i];
  const f32x2 ix = {me.px, me.tx}, iy = {me.py, me.ty}, iz = {me.pz, me.tz};
  float offA = 0.f, offB = 0.f, diagA = 0.f, diagB = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
  float *const cf_row = s_cf + (w * TS + lane) * CF_LD;     // phase 1: this lane's row of coefficients

  // one pair from its column atom: e^2 into acc, cf (x_i - x_j) into the row gradient; returns cf (0 for a dead row).
  // 3 packed subtractions + 3 packed fma + 2 v_rsq_f32 + 1 packed multiply + 3 scalar ops for the loss, 4 more for the
  // row gradient.  clamp_min(1e-30) of losses.py:252 is the 1e-30 the squares are added to: the same number bit for
  // bit unless two atoms are within 1e-11 A of each other, 1e-30 (as there) when they coincide, and no v_max per distance.
  auto pair_of = [&](const Col8 &c, float &acc) __attribute__((always_inline)) {
    const f32x2 dx = ix - (f32x2){c.px, c.tx}, dy = iy - (f32x2){c.py, c.ty}, dz = iz - (f32x2){c.pz, c.tz};  // (pred, true)
    f32x2 q = __builtin_elementwise_fma(dx, dx, (f32x2){1e-30f, 1e-30f});
    q = __builtin_elementwise_fma(dy, dy, q);
    q = __builtin_elementwise_fma(dz, dz, q);
    const float inv = __builtin_amdgcn_rsqf(q[0]), invt = __builtin_amdgcn_rsqf(q[1]);
    f32x2 dt = q * (f32x2){inv, invt};  // (d, tau)
    asm("" : "+v"(dt));  // keep the rounded products: no FMA contraction into e, so pred == true gives e == 0 exactly
    const float e = dt[0] - dt[1];
    acc = fmaf(e, e, acc);
    const float cf = e * inv;
    if (WITH_GRAD) {
      gx = fmaf(cf, dx[0], gx);
      gy = fmaf(cf, dy[0], gy);
      gz = fmaf(cf, dz[0], gz);
    }
    return cf;
  };
  // The column tile is staged in LDS by the strip's last wavefront (the one with the least work in the strip's first
  // tiles), tile J + 1 into the other buffer in front of the barrier that ends tile J, and read as broadcasts in batches of
  // U atoms: per batch the U independent chains, then the U coefficient stores (pair by pair the compiler strings the
  // chains - two transcendentals and a dozen dependent VALU instructions each - one behind the other).  (Measured
  // alternative, round 4: the column atoms as a double-buffered stream of scalar loads, coordinates as SGPR operands -
  // no LDS reads at all, and 10 % slower: profiles/r04/NOTES.md section 5.)
  constexpr int U = PT_DRMSD_UNROLL;
  static_assert(SUB % U == 0, "a sub-block of coefficient columns is a whole number of batches");
  struct Batch {
    Col8 c[U];
  };
  auto fetch = [&](int buf, int j) __attribute__((always_inline)) {
    Batch t;
#pragma unroll
    for (int u = 0; u < U; ++u) t.c[u] = s_col[buf][j + u];
    return t;
  };
  auto stage = [&](int J, int buf) __attribute__((always_inline)) {   // one wavefront's job: tile J -> s_col[buf]
    const int jj = J * TS + lane;
    s_col[buf][lane] = jj < n ? col8[jj] : Col8{0, 0, 0, 0, 0, 0, 0, 0};
  };
  // columns [j, j + U) of a tile whose live columns are [0, j1), backbone columns [0, jb): whole batches on one accumulator
  // without a test per column, the (at most two per tile) ragged ones column by column; cf_out = nullptr_t: keep nothing
  auto batch = [&](const Batch &t, int j, int jb, int j1, auto cf_out) __attribute__((always_inline)) {
    constexpr bool keep = WITH_GRAD && !std::is_same<decltype(cf_out), std::nullptr_t>::value;
    if (j + U <= jb || (j >= jb && j + U <= j1)) {
      float acc = 0.f, cf[U];   // (a batch sums its U squares first: a reference to one of two accumulators would put both in memory)
#pragma unroll
      for (int u = 0; u < U; ++u) cf[u] = pair_of(t.c[u], acc);
      const bool bb = j + U <= jb;   // (wavefront-uniform; written as a branch the compiler selects between the ADDRESSES of the two
      offA += bb ? acc : 0.f;        //  accumulators and keeps both in scratch memory)
      offB += bb ? 0.f : acc;
      if constexpr (keep) {
#pragma unroll
        for (int u = 0; u < U; ++u) cf_out[u] = cf[u];
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (j + u < j1) {
          float e2 = 0.f;
          const float cf = pair_of(t.c[u], e2);
          offA += j + u < jb ? e2 : 0.f;
          offB += j + u < jb ? 0.f : e2;
          if constexpr (keep) cf_out[u] = cf;
        }
      }
    }
  };

  if (w == STRIP_TILES - 1) stage(J0, 0);
  __syncthreads();
  for (int J = J0; J < J1; ++J) {
    const int cbuf = (J - J0) & 1;
    if (w == STRIP_TILES - 1 && J + 1 < J1) stage(J + 1, cbuf ^ 1);
    const int cnt = min(TS, n - J * TS);
    const int ja = max(0, min(cnt, nbb - J * TS));   // columns below ja are backbone atoms (then so is every row i < j)
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);   // (S, Vx, Vy, Vz) of column lane of the tile, in lanes 0..cnt-1
    if (I < J) {  // wavefront-uniform: a full tile above the diagonal
      const float *cf_q = s_cf + (w * TS + (lane >> 4) * SUB) * CF_LD + (lane & (SUB - 1));   // phase 2: this lane's quarter of the rows
      for (int j0 = 0; j0 < cnt; j0 += SUB) {
        const int j1 = min(cnt, j0 + SUB), jb = min(j1, ja);
#pragma unroll
        for (int k = 0; k < SUB / U; ++k) {
          batch(fetch(cbuf, j0 + k * U), j0 + k * U, jb, j1, cf_row + k * U);
        }
        if (WITH_GRAD) {
          // phase 2 for these 16 columns (LDS is in order per wavefront): lane = (column j0 + (lane & 15), quarter lane >> 4
          // of the rows); the coordinates of row 16 quarter + r live in lane r of this lane's row of 16 lanes.  Every lane
          // takes part (a DPP operand of a disabled lane is not readable): columns >= j1 sum stale tile contents and are
          // dropped below - a lane only ever meets lanes of its own column
          float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
          colsum_rows(q, cf_q, me.px, me.py, me.pz, std::make_integer_sequence<int, SUB>{});
          // fold the four row quarters (lanes l, l ^ 16, l ^ 32, l ^ 48) in a fixed order: every lane ends with the sum
          q.x += __shfl_xor(q.x, 16, 64); q.y += __shfl_xor(q.y, 16, 64); q.z += __shfl_xor(q.z, 16, 64); q.w += __shfl_xor(q.w, 16, 64);
          q.x += __shfl_xor(q.x, 32, 64); q.y += __shfl_xor(q.y, 32, 64); q.z += __shfl_xor(q.z, 32, 64); q.w += __shfl_xor(q.w, 32, 64);
          // lane l keeps column l of the tile: quarter j0 / 16 holds columns j0 .. j0 + 15
          if ((lane >> 4) == (j0 >> 4) && j0 + (lane & (SUB - 1)) < j1) cs = q;
        }
      }
    } else if (I == J) {  // the diagonal tile: both sides inside the tile, j == i contributes exactly 0
      float &accA = diagA, &accB = diagB;
      auto dbatch = [&](const Batch &t, int j) __attribute__((always_inline)) {
        if (j + U <= ja || (j >= ja && j + U <= cnt)) {
          float acc = 0.f;
#pragma unroll
          for (int u = 0; u < U; ++u) pair_of(t.c[u], acc);
          accA += j + U <= ja ? acc : 0.f;
          accB += j + U <= ja ? 0.f : acc;
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (j + u < cnt) {
              float e2 = 0.f;
              pair_of(t.c[u], e2);
              accA += j + u < ja ? e2 : 0.f;
              accB += j + u < ja ? 0.f : e2;
            }
        }
      };
      for (int j = 0; j < cnt; j += U) dbatch(fetch(cbuf, j), j);
    }
    if (WITH_GRAD) {
      // two sets of slots, alternating: wavefront 0 reads set p behind this barrier while the others fill set p ^ 1 for
      // the next column tile and meet it at the next barrier - one barrier per column tile
      float4 *const slots = s_cs + cbuf * RS;
      slots[w * TS + lane] = cs;
      __syncthreads();
      if (w == 0) {  // fixed order over the four wavefronts
        const float4 a0 = slots[lane], a1 = slots[TS + lane], a2 = slots[2 * TS + lane], a3 = slots[3 * TS + lane];
        const float4 t = make_float4(((a0.x + a1.x) + a2.x) + a3.x, ((a0.y + a1.y) + a2.y) + a3.y,
                                     ((a0.z + a1.z) + a2.z) + a3.z, ((a0.w + a1.w) + a2.w) + a3.w);
        colpart[(((size_t)b * spp + sl) * tl.tiles + J) * TS + lane] = t;
      }
    } else {
      __syncthreads();   // the next column tile is staged; this one is free
    }
  }
  if (WITH_GRAD) rowpart[(((size_t)b * spp + sl) * tl.chunks + chunk) * RS + tid] = make_float4(gx, gy, gz, 0.f);
  // loss terms: off-diagonal tiles saw every pair once, the diagonal tile twice
  double all = (double)offA + (double)offB + 0.5 * ((double)diagA + (double)diagB);
  double bbp = (double)offA + ((live && i < nbb) ? 0.5 * (double)diagA : 0.0);
  all = wave_sum_d(all);
  bbp = wave_sum_d(bbp);
  if (lane == 0) {
    s_red[w * 2] = all;
    s_red[w * 2 + 1] = bbp;
  }
  __syncthreads();
  if (tid == 0) {
    double a = 0, c = 0;
    for (int k = 0; k < STRIP_TILES; ++k) {
      a += s_red[k * 2];
      c += s_red[k * 2 + 1];
    }
    part[0] = a;
    part[1] = c;
  }
}
constexpr size_t TRI_LDS = (size_t)(STRIP_TILES * TS * CF_LD) * 4 + 2 * RS * 16;

// The gradient of atom j from the partial sums of ONE launch of the pair sweep over the strips strip0 .. strip0 + spp - 1:
//   row side     the chunks of the atom's own strip that had work - a contiguous range - in chunk order (only when that strip is
//                in the launch);
//   column side  g += x_j S - V for every strip of the launch at or above the atom's tile, in strip order, continuing `col`.
// Loads go out four at a time (a step of few proteins runs these kernels on a fraction of the CUs: dependent-looking loads
// were their time).  The two sides are summed SEPARATELY (round 5) so that a long batch can be swept in passes over groups of
// strips - the column sums of pass k continue those of pass k - 1 - and give the same bits as one launch.
struct RowCol {
  float rx, ry, rz, cx, cy, cz;
};
__device__ __forceinline__ void gather_partials(const TriLayout &tl, int b, int j, int nT, const float4 &xj,
                                                const float4 *__restrict__ rowpart, const float4 *__restrict__ colpart,
                                                int strip0, int spp, RowCol &g) {
  const int J = j / TS, sJ = J / STRIP_TILES;
  if (sJ >= strip0 && sJ < strip0 + spp) {
    const int c0 = (STRIP_TILES * sJ) / tl.chunk_tiles, c1 = min(tl.chunks, (nT + tl.chunk_tiles - 1) / tl.chunk_tiles);
    const float4 *rp = rowpart + (((size_t)b * spp + (sJ - strip0)) * tl.chunks) * RS + (j - sJ * RS);
    int c = c0;
    for (; c + 4 <= c1; c += 4) {
      const float4 r0 = rp[(size_t)c * RS], r1 = rp[(size_t)(c + 1) * RS], r2 = rp[(size_t)(c + 2) * RS], r3 = rp[(size_t)(c + 3) * RS];
      g.rx += r0.x; g.ry += r0.y; g.rz += r0.z;
      g.rx += r1.x; g.ry += r1.y; g.rz += r1.z;
      g.rx += r2.x; g.ry += r2.y; g.rz += r2.z;
      g.rx += r3.x; g.ry += r3.y; g.rz += r3.z;
    }
    for (; c < c1; ++c) {
      const float4 r = rp[(size_t)c * RS];
      g.rx += r.x; g.ry += r.y; g.rz += r.z;
    }
  }
  const int last = min(sJ, strip0 + spp - 1) - strip0;     // strips strip0 .. strip0 + last of this launch lie at or above tile J
  if (last >= 0) {
    const float4 *cp_ = colpart + ((size_t)b * spp * tl.tiles + J) * TS + (j & (TS - 1));
    const size_t step = (size_t)tl.tiles * TS;
    auto add = [&](const float4 &cp) __attribute__((always_inline)) {
      g.cx += fmaf(xj.x, cp.x, -cp.y);
      g.cy += fmaf(xj.y, cp.x, -cp.z);
      g.cz += fmaf(xj.z, cp.x, -cp.w);
    };
    int sidx = 0;
    for (; sidx + 4 <= last + 1; sidx += 4) {
      const float4 q0 = cp_[sidx * step], q1 = cp_[(sidx + 1) * step], q2 = cp_[(sidx + 2) * step], q3 = cp_[(sidx + 3) * step];
      add(q0); add(q1); add(q2); add(q3);
    }
    for (; sidx <= last; ++sidx) add(cp_[sidx * step]);
  }
}

// between the passes of a long batch: the running (row sum, column sum) of every atom, grid (ceil(nmax / 256), B)
__global__ __launch_bounds__(CB) void drmsd_accumulate_kernel(const Counts *__restrict__ counts, const float4 *__restrict__ pred4,
                                                              const float4 *__restrict__ rowpart, const float4 *__restrict__ colpart,
                                                              int L, int strip0, int spp, float4 *__restrict__ grow,
                                                              float4 *__restrict__ gcol) {
  const int b = blockIdx.y, j = blockIdx.x * CB + threadIdx.x;
  const Counts cn = counts[b];
  if (j >= cn.n) return;
  const size_t nmax = (size_t)L * 14;
  const TriLayout tl = tri_layout((int)nmax, (int)gridDim.y);
  const size_t at = (size_t)b * nmax + j;
  RowCol g = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (strip0 > 0) {
    const float4 r = grow[at], c = gcol[at];
    g = RowCol{r.x, r.y, r.z, c.x, c.y, c.z};
  }
  gather_partials(tl, b, j, (cn.n + TS - 1) / TS, pred4[at], rowpart, colpart, strip0, spp, g);
  grow[at] = make_float4(g.rx, g.ry, g.rz, 0.f);
  gcol[at] = make_float4(g.cx, g.cy, g.cz, 0.f);
}

// statistics, gradient assembly and scatter back to the slot layout: grid (ceil(nmax / 256), B); dcrd was zeroed before.
// grow / gcol != nullptr: the sweep ran in passes and drmsd_accumulate_kernel has already summed the partials.
__global__ __launch_bounds__(CB) void drmsd_finalize_kernel(const Counts *__restrict__ counts,
                                                            const double *__restrict__ partials,
                                                            const float4 *__restrict__ pred4,
                                                            const float4 *__restrict__ rowpart,
                                                            const float4 *__restrict__ colpart,
                                                            const float4 *__restrict__ grow, const float4 *__restrict__ gcol,
                                                            const int *__restrict__ idx, int L,
                                                            float *__restrict__ stats, float *__restrict__ dcrd) {
  __shared__ float s_scale;
  const int b = blockIdx.y, tid = threadIdx.x;
  const Counts cn = counts[b];
  const size_t nmax = (size_t)L * 14;
  const TriLayout tl = tri_layout((int)nmax, (int)gridDim.y);
  if ((int)blockIdx.x * CB >= max(cn.n, 1)) return;
  // the work items' loss partials: every thread adds its share in item order, then a fixed tree over the threads (every
  // block of the protein needs the total for the gradient scale: one thread walking strips x chunks items was the
  // kernel's time when small chunks make that 784 items)
  __shared__ double s_sum[2][CB];
  {
    double a = 0, c = 0;
    const int items = tl.strips * tl.chunks;
    for (int r = tid; r < items; r += CB) {
      a += partials[((size_t)b * items + r) * 2];
      c += partials[((size_t)b * items + r) * 2 + 1];
    }
    s_sum[0][tid] = a;
    s_sum[1][tid] = c;
    __syncthreads();
    for (int o = CB / 2; o > 0; o >>= 1) {
      if (tid < o) {
        s_sum[0][tid] += s_sum[0][tid + o];
        s_sum[1][tid] += s_sum[1][tid + o];
      }
      __syncthreads();
    }
  }
  if (tid == 0) {
    const double all = s_sum[0][0], bbp = s_sum[1][0];
    const double n = cn.n, nb = cn.n_bb;
    const double P = n * (n - 1) * 0.5, Pb = nb * (nb - 1) * 0.5;
    // mse_loss over an empty pair set is NaN in the reference as well
    const float D = (float)sqrt(all / P);
    const float Db = (float)sqrt(bbp / Pb);
    if (blockIdx.x == 0) {
      float *st = stats + (size_t)b * 8;
      st[0] = D;
      st[1] = D / (float)cn.n;
      st[2] = Db;
      st[3] = Db / (float)cn.n_bb;
      st[4] = (float)cn.n;
      st[5] = (float)cn.n_bb;
      st[6] = 0.f;
      st[7] = 0.f;
    }
    s_scale = (float)(1.0 / (n * P * (double)D));
  }
  if (dcrd == nullptr) return;
  __syncthreads();
  const int j = blockIdx.x * CB + tid;
  if (j >= cn.n) return;
  const float scale = s_scale;
  const size_t at = (size_t)b * nmax + j;
  RowCol g = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (grow) {
    const float4 r = grow[at], c = gcol[at];
    g = RowCol{r.x, r.y, r.z, c.x, c.y, c.z};
  } else {
    gather_partials(tl, b, j, (cn.n + TS - 1) / TS, pred4[at], rowpart, colpart, 0, tl.strips, g);
  }
  const int sl = idx[at];
  float *out = dcrd + (size_t)b * nmax * 3;
  out[sl * 3 + 0] = scale * (g.rx + g.cx);
  out[sl * 3 + 1] = scale * (g.ry + g.cy);
  out[sl * 3 + 2] = scale * (g.rz + g.cz);
}

// The fixed-order partial sums of the sweep grow as O(n^2 / 256) per protein: 166 MB at (32, 512), 1.35 GB at (32, 1500).
// Beyond PARTIAL_BUDGET the strips are swept in PASSES of `spp` strips (one launch of the pair kernel + one of
// drmsd_accumulate_kernel each, same buffers): the workspace stays below ~256 MB at any size, same bits.  A function of
// (B, L) and of the budget alone: the plain entry points use PARTIAL_BUDGET, the *_budget ones take it as an ARGUMENT (tests
// force passes on a small batch with it) - nothing is read from the process environment.
constexpr size_t PARTIAL_BUDGET = (size_t)200 << 20;
struct Layout {
  size_t pred4, col8, rowpart, colpart, grow, gcol, idx, counts, partials, total;
  int spp;   // strips per pass (= strips: one launch)
  TriLayout tl;
};
Layout layout(int B, int L, size_t budget) {
  Layout l;
  const size_t nmax = (size_t)L * 14, BN = (size_t)B * nmax;
  l.tl = tri_layout((int)nmax, B);
  if (budget == 0) budget = PARTIAL_BUDGET;
  const size_t per_strip = (size_t)B * ((size_t)l.tl.chunks * RS + (size_t)l.tl.tiles * TS) * sizeof(float4);
  l.spp = l.tl.strips;
  if (per_strip * l.tl.strips > budget) l.spp = (int)max((size_t)1, budget / per_strip);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  l.pred4 = take(BN * sizeof(float4));
  l.col8 = take(BN * sizeof(Col8));
  l.rowpart = take((size_t)B * l.spp * l.tl.chunks * RS * sizeof(float4));
  l.colpart = take((size_t)B * l.spp * l.tl.tiles * TS * sizeof(float4));
  const bool passes = l.spp < l.tl.strips;
  l.grow = take(passes ? BN * sizeof(float4) : 0);
  l.gcol = take(passes ? BN * sizeof(float4) : 0);
  l.idx = take(BN * sizeof(int));
  l.counts = take((size_t)B * sizeof(Counts));
  l.partials = take((size_t)B * l.tl.strips * l.tl.chunks * 2 * sizeof(double));
  l.total = off;
  return l;
}

}  // namespace

extern "C" {

size_t ptamd_drmsd_workspace_bytes_budget(int B, int L, size_t partial_budget_bytes) {
  if (B <= 0 || L <= 0) return 0;
  return layout(B, L, partial_budget_bytes).total;
}
size_t ptamd_drmsd_workspace_bytes(int B, int L) { return ptamd_drmsd_workspace_bytes_budget(B, L, 0); }

int ptamd_drmsd_fwd_bwd(const float *pred_crd, const float *true_crd, const int64_t *seq, int B, int L, float *stats,
                        float *dcrd, void *workspace, size_t workspace_bytes, void *stream) {
  return ptamd_drmsd_fwd_bwd_budget(pred_crd, true_crd, seq, B, L, stats, dcrd, workspace, workspace_bytes, 0, stream);
}

int ptamd_drmsd_fwd_bwd_budget(const float *pred_crd, const float *true_crd, const int64_t *seq, int B, int L, float *stats,
                               float *dcrd, void *workspace, size_t workspace_bytes, size_t partial_budget_bytes, void *stream) {
  if (B <= 0 || L <= 0) return PTAMD_ERR_BAD_SHAPE;
  const Layout l = layout(B, L, partial_budget_bytes);
  if (!workspace || workspace_bytes < l.total) return PTAMD_ERR_WORKSPACE;
  if (!pt_aligned16(workspace)) return PTAMD_ERR_ALIGN;
  char *ws = static_cast<char *>(workspace);
  Col8 *col8 = reinterpret_cast<Col8 *>(ws + l.col8);
  float4 *pred4 = reinterpret_cast<float4 *>(ws + l.pred4),
         *rowpart = reinterpret_cast<float4 *>(ws + l.rowpart), *colpart = reinterpret_cast<float4 *>(ws + l.colpart);
  int *idx = reinterpret_cast<int *>(ws + l.idx);
  Counts *counts = reinterpret_cast<Counts *>(ws + l.counts);
  double *partials = reinterpret_cast<double *>(ws + l.partials);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(drmsd_compact_kernel, dim3(B), dim3(COMPACT_THREADS), 0, st, pred_crd, true_crd, seq, L, pred4, col8, idx,
                     counts);
  int rc = pt_check_launch();
  if (rc) return rc;
  const bool passes = l.spp < l.tl.strips;
  float4 *grow = passes ? reinterpret_cast<float4 *>(ws + l.grow) : nullptr, *gcol = passes ? reinterpret_cast<float4 *>(ws + l.gcol) : nullptr;
  const dim3 fin_grid((unsigned)(((size_t)L * 14 + CB - 1) / CB), B);
  if (dcrd) PT_HIP_TRY(hipMemsetAsync(dcrd, 0, (size_t)B * L * 14 * 3 * sizeof(float), st));   // slots of absent atoms stay 0
  auto kern = dcrd ? drmsd_tri_kernel<true> : drmsd_tri_kernel<false>;
  PT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TRI_LDS));
  for (int s0 = 0; s0 < l.tl.strips; s0 += l.spp) {
    const int ns = l.tl.strips - s0 < l.spp ? l.tl.strips - s0 : l.spp;
    hipLaunchKernelGGL(kern, dim3(ns * l.tl.chunks, B), dim3(RS), TRI_LDS, st, col8, counts, L, rowpart, colpart, partials, s0, l.spp);
    rc = pt_check_launch();
    if (rc) return rc;
    if (passes && dcrd) {
      hipLaunchKernelGGL(drmsd_accumulate_kernel, fin_grid, dim3(CB), 0, st, counts, pred4, rowpart, colpart, L, s0, l.spp,
                         grow, gcol);
      rc = pt_check_launch();
      if (rc) return rc;
    }
  }
  hipLaunchKernelGGL(drmsd_finalize_kernel, fin_grid, dim3(CB), 0, st, counts, partials, pred4, rowpart, colpart,
                     dcrd ? grow : nullptr, dcrd ? gcol : nullptr, idx, L, stats, dcrd);
  return pt_check_launch();
}

}  // extern "C"
