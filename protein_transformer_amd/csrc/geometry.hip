// Angle recovery (atan2) and the NeRF all-atom build, forward and adjoint, for gfx950.
//
// Replaces the per-atom Python/PyTorch op chain of
//   nerf                      /root/reference/protein_transformer/protein/Structure.py:23-65
//   StructureBuilder.build    .../protein/StructureBuilder.py:55-92
//   ResidueBuilder.build_bb   .../protein/StructureBuilder.py:147-191
//   ResidueBuilder.build_sc   .../protein/StructureBuilder.py:193-236
//   inverse_trig_transform    .../losses.py:26-36
// and the autograd graph the reference builds over them (losses.py:91-92).
//
// Structure of the work (one protein = one dependent chain of 3L backbone placements):
//   nerf_backbone_*  one wavefront per protein; all lanes stage sin/cos (and, backward, coordinates and
//                    incoming adjoints) into LDS in parallel, then lane 0 walks the chain out of LDS with
//                    the next residue's operands prefetched.  Latency-bound by construction.
//   nerf_sidechain_* one lane per residue: O plus up to 10 table-driven side-chain placements, per-lane
//                    atom arrays live in LDS ([slot][xyz][lane], conflict-free) because parent slots are
//                    data dependent.
#include "common.h"
#include "nerf_tables.h"

namespace {

constexpr float BL_N_CA = 1.442f, BL_CA_C = 1.498f, BL_C_N = 1.379f, BL_C_O = 1.229f, BA_CA_C_O = 2.0944f;
constexpr float PI_F = 3.141592653589793f;
constexpr int MAX_L_CHAIN = 1024;  // LDS staging: 144 B per residue in the backward walk

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// torch.nn.functional.normalize: v / max(||v||, 1e-12)
__device__ __forceinline__ V3 unit(V3 v, float &nrm) {
  nrm = fmaxf(sqrtf(dot(v, v)), 1e-12f);
  return {v.x / nrm, v.y / nrm, v.z / nrm};
}

// Structure.py:44-65 with W_hat supplied (it equals the previous placement's x_hat on the backbone).
__device__ __forceinline__ V3 place_with_w(V3 W, V3 b, V3 c, float l, float st, float ct, float sx, float cx,
                                           V3 &x_out) {
  float nw, nn;
  V3 x = unit(c - b, nw);
  V3 z = unit(cross(W, x), nn);
  V3 y = cross(z, x);
  float d0 = -l * ct, d1 = l * st * cx, d2 = l * st * sx;
  x_out = x;
  return {c.x + (x.x * d0 + y.x * d1 + z.x * d2), c.y + (x.y * d0 + y.y * d1 + z.y * d2),
          c.z + (x.z * d0 + y.z * d1 + z.z * d2)};
}
__device__ __forceinline__ V3 place(V3 a, V3 b, V3 c, float l, float st, float ct, float sx, float cx) {
  float nu;
  V3 W = unit(b - a, nu), xo;
  return place_with_w(W, b, c, l, st, ct, sx, cx, xo);
}

// Adjoint of one placement.  g = dL/dd; accumulates into ga, gb, gc, gth, gchi.
__device__ __forceinline__ void place_bwd(V3 a, V3 b, V3 c, float l, float st, float ct, float sx, float cx, V3 g,
                                          V3 &ga, V3 &gb, V3 &gc, float &gth, float &gchi) {
  float nu, nw, nn;
  V3 u = b - a, w = c - b;
  float ru = sqrtf(dot(u, u)), rw = sqrtf(dot(w, w));
  V3 W = unit(u, nu), x = unit(w, nw);
  V3 n = cross(W, x);
  float rn = sqrtf(dot(n, n));
  V3 z = unit(n, nn);
  V3 y = cross(z, x);
  float v0 = -l * ct, v1 = l * st * cx, v2 = l * st * sx;
  gc = gc + g;
  V3 xb = v0 * g, yb = v1 * g, zb = v2 * g;
  float vb0 = dot(g, x), vb1 = dot(g, y), vb2 = dot(g, z);
  gth += l * (st * vb0 + ct * cx * vb1 + ct * sx * vb2);
  gchi += l * st * (-sx * vb1 + cx * vb2);
  zb = zb + cross(x, yb);
  xb = xb + cross(yb, z);
  V3 nb = (rn > 1e-12f) ? (1.f / nn) * (zb - dot(zb, z) * z) : (1.f / nn) * zb;
  V3 Wb = cross(x, nb);
  xb = xb + cross(nb, W);
  V3 wb = (rw > 1e-12f) ? (1.f / nw) * (xb - dot(xb, x) * x) : (1.f / nw) * xb;
  gc = gc + wb;
  gb = gb - wb;
  V3 ub = (ru > 1e-12f) ? (1.f / nu) * (Wb - dot(Wb, W) * W) : (1.f / nu) * Wb;
  gb = gb + ub;
  ga = ga - ub;
}

__device__ __forceinline__ V3 ld3(const float *p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float *p, V3 v) {
  p[0] = v.x;
  p[1] = v.y;
  p[2] = v.z;
}

// number of leading non-pad residues of one protein; flags ids outside 0..19 (wave-cooperative)
__device__ int protein_len(const int64_t *seq, int L, int lane, int32_t *status) {
  int cnt = 0, bad = 0;
  for (int i = lane; i < L; i += PT_WAVE) {
    int64_t r = seq[i];
    if (r != PTAMD_PAD_ID) {
      ++cnt;
      if (r < 0 || r > 19) bad = 1;
    }
  }
  cnt = (int)wave_sum((float)cnt);
  if (__any(bad) && lane == 0) atomicOr(status, PTAMD_ST_BAD_RESIDUE);
  return cnt;
}

// ------------------------------------------------------------------------------------------------
// angles
__global__ void angles_fwd_kernel(const float2 *__restrict__ sc, float *__restrict__ ang, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float2 v = sc[i];
    ang[i] = atan2f(v.y, v.x);
  }
}
__global__ void angles_bwd_kernel(const float2 *__restrict__ sc, const float *__restrict__ dang,
                                  float2 *__restrict__ dsc, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float2 v = sc[i];
    float g = dang[i], r2 = v.x * v.x + v.y * v.y;
    float inv = r2 > 0.f ? 1.f / r2 : 0.f;
    dsc[i] = make_float2(-v.y * inv * g, v.x * inv * g);
  }
}

// ------------------------------------------------------------------------------------------------
// backbone chain, forward.  grid = B, block = 64.  LDS: trig[L][12] = (sin,cos) of angle columns 0..5
__global__ __launch_bounds__(PT_WAVE) void nerf_backbone_fwd_kernel(const float *__restrict__ ang,
                                                                    const int64_t *__restrict__ seq, int L,
                                                                    float *__restrict__ crd,
                                                                    int32_t *__restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.x, lane = threadIdx.x;
  ang += (size_t)b * L * 12;
  seq += (size_t)b * L;
  crd += (size_t)b * L * 42;
  const int len = protein_len(seq, L, lane, status);
  if (len < 2) {
    if (lane == 0) atomicOr(status, PTAMD_ST_TOO_SHORT);
    return;  // the side-chain kernel zero-fills
  }
  int bad_theta = 0;
  for (int t = lane; t < len * 6; t += PT_WAVE) {
    int i = t / 6, k = t - i * 6;
    float a = ang[i * 12 + k], s, c;
    sincosf(a, &s, &c);
    lds[i * 12 + 2 * k] = s;
    lds[i * 12 + 2 * k + 1] = c;
    if (k >= 3 && !(fabsf(a) <= PI_F)) bad_theta = 1;  // fp32 pi itself is accepted
  }
  if (__any(bad_theta) && lane == 0) atomicOr(status, PTAMD_ST_BAD_THETA);
  __syncthreads();
  if (lane != 0) return;

  // init_bb (StructureBuilder.py:181-191)
  V3 pN = {0.f, 0.f, 0.001f};
  V3 pCA = {pN.x + BL_N_CA, pN.y + 0.f, pN.z + 0.f};
  float a03 = ang[3];
  V3 pC = {pCA.x + cosf(PI_F - a03) * BL_CA_C, pCA.y + sinf(PI_F - a03) * BL_CA_C, pCA.z + 0.f};
  st3(crd + 0, pN);
  st3(crd + 3, pCA);
  st3(crd + 6, pC);
  float nrm;
  V3 W = unit(pCA - pN, nrm);  // x_hat of the (virtual) CA placement = W_hat for the next C-type placement
  V3 Wn = unit(pC - pCA, nrm);
  // W for N_1 is unit(CA_0 - N_0); after each placement the new x_hat becomes the W of the next one.
  const float4 *tr = reinterpret_cast<const float4 *>(lds);
  float4 p0 = tr[0], p1 = tr[1], p2 = tr[2];  // residue i-1: (s0,c0,s1,c1) (s2,c2,s3,c3) (s4,c4,s5,c5)
  for (int i = 1; i < len; ++i) {
    float4 q0 = tr[i * 3], q1 = tr[i * 3 + 1], q2 = tr[i * 3 + 2];
    V3 xo;
    // N_i = nerf(N_{i-1}, CA_{i-1}, C_{i-1}; c-n, theta = prev[4], chi = prev[1])
    V3 N = place_with_w(W, pCA, pC, BL_C_N, p2.x, p2.y, p0.z, p0.w, xo);
    // CA_i = nerf(CA_{i-1}, C_{i-1}, N_i; n-ca, theta = prev[5], chi = prev[2])
    V3 Wca = xo;  // = unit(C_{i-1} - CA_{i-1}) recomputed inside the call above as x_hat
    V3 xo2;
    V3 CA = place_with_w(Wca, pC, N, BL_N_CA, p2.z, p2.w, p1.x, p1.y, xo2);
    // C_i = nerf(C_{i-1}, N_i, CA_i; ca-c, theta = cur[3], chi = cur[0])
    V3 xo3;
    V3 C = place_with_w(xo2, N, CA, BL_CA_C, q1.z, q1.w, q0.x, q0.y, xo3);
    st3(crd + i * 42 + 0, N);
    st3(crd + i * 42 + 3, CA);
    st3(crd + i * 42 + 6, C);
    W = xo3;  // unit(CA_i - N_i): W_hat of N_{i+1}
    pN = N;
    pCA = CA;
    pC = C;
    p0 = q0;
    p1 = q1;
    p2 = q2;
    (void)Wn;
  }
}

// ------------------------------------------------------------------------------------------------
// O + side chain, forward. one lane per residue; grid.x = ceil(L/64), grid.y = B
constexpr int SC_BLOCK = 64;
#define PTS(slot, comp) pts[((slot) * 3 + (comp)) * SC_BLOCK + tid]
#define GRD(slot, comp) grd[((slot) * 3 + (comp)) * SC_BLOCK + tid]

__device__ __forceinline__ V3 lds_get(const float *pts, int slot, int tid) {
  return {PTS(slot, 0), PTS(slot, 1), PTS(slot, 2)};
}
__device__ __forceinline__ void lds_put(float *pts, int slot, int tid, V3 v) {
  PTS(slot, 0) = v.x;
  PTS(slot, 1) = v.y;
  PTS(slot, 2) = v.z;
}

__global__ __launch_bounds__(SC_BLOCK) void nerf_sidechain_fwd_kernel(const float *__restrict__ ang,
                                                                      const int64_t *__restrict__ seq, int L,
                                                                      float *__restrict__ crd) {
  __shared__ float pts[14 * 3 * SC_BLOCK];
  __shared__ int s_len;
  const int b = blockIdx.y, tid = threadIdx.x, i = blockIdx.x * SC_BLOCK + tid;
  ang += (size_t)b * L * 12;
  seq += (size_t)b * L;
  crd += (size_t)b * L * 42;
  {
    int32_t dummy = 0;
    int len = protein_len(seq, L, tid, &dummy);
    if (tid == 0) s_len = len;
  }
  __syncthreads();
  const int len = s_len;
  if (i >= L) return;
  float *out = crd + (size_t)i * 42;
  if (i >= len || len < 2) {
    for (int k = 0; k < 42; ++k) out[k] = 0.f;
    return;
  }
  int res = (int)seq[i];
  if (res < 0 || res > 19) res = 5;  // flagged by the backbone kernel; build it as GLY
  const float *a = ang + (size_t)i * 12;
  V3 N = ld3(out), CA = ld3(out + 3), C = ld3(out + 6);
  lds_put(pts, 0, tid, N);
  lds_put(pts, 1, tid, CA);
  lds_put(pts, 2, tid, C);
  float s, c, st, ct;
  // O = nerf(N, CA, C; c-o, 2.0944, psi - pi)   (StructureBuilder.py:170-173,188-190)
  sincosf(BA_CA_C_O, &st, &ct);
  sincosf(a[1] - PI_F, &s, &c);
  st3(out + 9, place(N, CA, C, BL_C_O, st, ct, s, c));
  V3 ext = (i == 0) ? ld3(out + 42) /* N of residue 1 */ : ld3(out - 42 + 6) /* C of residue i-1 */;
  const int nsc = c_pt_nsc[res];
  float last = 0.f;
  for (int k = 0; k < nsc; ++k) {
    const PtScAtom at = c_pt_sc[res][k];
    V3 pa, pb, pc;
    if (k == 0) {
      if (i == 0) {
        pa = ext; pb = C; pc = CA;
      } else {
        pa = ext; pb = N; pc = CA;
      }
    } else {
      pa = lds_get(pts, at.pa, tid);
      pb = lds_get(pts, at.pb, tid);
      pc = lds_get(pts, at.pc, tid);
    }
    float chi = at.kind == PT_TORS_PRED ? a[6 + (k < 6 ? k : 5)] : (at.kind == PT_TORS_INFER ? last - PI_F : at.tors_const);
    sincosf(at.angle, &st, &ct);
    sincosf(chi, &s, &c);
    V3 d = place(pa, pb, pc, at.bond, st, ct, s, c);
    lds_put(pts, 4 + k, tid, d);
    st3(out + (4 + k) * 3, d);
    last = chi;
  }
  for (int k = (4 + nsc) * 3; k < 42; ++k) out[k] = 0.f;
}

// ------------------------------------------------------------------------------------------------
// O + side chain, adjoint.  Writes dang[b,i,:] (chi columns, psi from O, zeros elsewhere) and
// gbb[b,i,0..11] = adjoints flowing into (N, CA, C, ext) where ext = C_{i-1} (i>=1) or N_1 (i==0).
__global__ __launch_bounds__(SC_BLOCK) void nerf_sidechain_bwd_kernel(const float *__restrict__ ang,
                                                                      const int64_t *__restrict__ seq,
                                                                      const float *__restrict__ crd,
                                                                      const float *__restrict__ dcrd, int L,
                                                                      float *__restrict__ dang,
                                                                      float *__restrict__ gbb) {
  __shared__ float pts[14 * 3 * SC_BLOCK];
  __shared__ float grd[14 * 3 * SC_BLOCK];
  __shared__ int s_len;
  const int b = blockIdx.y, tid = threadIdx.x, i = blockIdx.x * SC_BLOCK + tid;
  ang += (size_t)b * L * 12;
  seq += (size_t)b * L;
  crd += (size_t)b * L * 42;
  dcrd += (size_t)b * L * 42;
  dang += (size_t)b * L * 12;
  gbb += (size_t)b * L * 12;
  {
    int32_t dummy = 0;
    int len = protein_len(seq, L, tid, &dummy);
    if (tid == 0) s_len = len;
  }
  __syncthreads();
  const int len = s_len;
  if (i >= L) return;
  float *da = dang + (size_t)i * 12;
  float *gb = gbb + (size_t)i * 12;
  if (i >= len || len < 2) {
    for (int k = 0; k < 12; ++k) {
      da[k] = 0.f;
      gb[k] = 0.f;
    }
    return;
  }
  int res = (int)seq[i];
  if (res < 0 || res > 19) res = 5;
  const float *a = ang + (size_t)i * 12;
  const float *p = crd + (size_t)i * 42;
  const float *g = dcrd + (size_t)i * 42;
  const int nsc = c_pt_nsc[res];
  for (int s = 0; s < 4 + nsc; ++s) {
    lds_put(pts, s, tid, ld3(p + s * 3));
    lds_put(grd, s, tid, ld3(g + s * 3));
  }
  V3 ext = (i == 0) ? ld3(p + 42) : ld3(p - 42 + 6);
  V3 gext = {0.f, 0.f, 0.f};
  float dchi[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // chi of every atom is needed before walking backwards ('i' atoms depend on the previous torsion)
  for (int k = nsc - 1; k >= 0; --k) {
    const PtScAtom at = c_pt_sc[res][k];
    float chi;
    int src;  // angle column that receives d/dchi, -1 for constants
    if (at.kind == PT_TORS_PRED) {
      chi = a[6 + (k < 6 ? k : 5)];
      src = k;
    } else if (at.kind == PT_TORS_INFER) {
      const PtScAtom pv = c_pt_sc[res][k - 1];  // planar partners always follow a predicted torsion
      chi = (pv.kind == PT_TORS_PRED ? a[6 + (k - 1 < 6 ? k - 1 : 5)] : pv.tors_const) - PI_F;
      src = pv.kind == PT_TORS_PRED ? k - 1 : -1;
    } else {
      chi = at.tors_const;
      src = -1;
    }
    float st, ct, s, c;
    sincosf(at.angle, &st, &ct);
    sincosf(chi, &s, &c);
    V3 gd = lds_get(grd, 4 + k, tid);
    V3 ga = {0, 0, 0}, gbv = {0, 0, 0}, gc = {0, 0, 0};
    float gth = 0.f, gchi = 0.f;
    if (k == 0) {
      V3 pb = (i == 0) ? lds_get(pts, 2, tid) : lds_get(pts, 0, tid);
      V3 pc = lds_get(pts, 1, tid);
      place_bwd(ext, pb, pc, at.bond, st, ct, s, c, gd, ga, gbv, gc, gth, gchi);
      gext = gext + ga;
      int sb = (i == 0) ? 2 : 0;
      lds_put(grd, sb, tid, lds_get(grd, sb, tid) + gbv);
      lds_put(grd, 1, tid, lds_get(grd, 1, tid) + gc);
    } else {
      place_bwd(lds_get(pts, at.pa, tid), lds_get(pts, at.pb, tid), lds_get(pts, at.pc, tid), at.bond, st, ct, s, c,
                gd, ga, gbv, gc, gth, gchi);
      lds_put(grd, at.pa, tid, lds_get(grd, at.pa, tid) + ga);
      lds_put(grd, at.pb, tid, lds_get(grd, at.pb, tid) + gbv);
      lds_put(grd, at.pc, tid, lds_get(grd, at.pc, tid) + gc);
    }
    if (src >= 0 && src < 6) dchi[src] += gchi;
  }
  // O
  float dpsi = 0.f;
  {
    float st, ct, s, c, gth = 0.f;
    sincosf(BA_CA_C_O, &st, &ct);
    sincosf(a[1] - PI_F, &s, &c);
    V3 ga = {0, 0, 0}, gbv = {0, 0, 0}, gc = {0, 0, 0};
    place_bwd(lds_get(pts, 0, tid), lds_get(pts, 1, tid), lds_get(pts, 2, tid), BL_C_O, st, ct, s, c,
              lds_get(grd, 3, tid), ga, gbv, gc, gth, dpsi);
    lds_put(grd, 0, tid, lds_get(grd, 0, tid) + ga);
    lds_put(grd, 1, tid, lds_get(grd, 1, tid) + gbv);
    lds_put(grd, 2, tid, lds_get(grd, 2, tid) + gc);
  }
  da[0] = 0.f;
  da[1] = dpsi;
  da[2] = da[3] = da[4] = da[5] = 0.f;
  for (int k = 0; k < 6; ++k) da[6 + k] = dchi[k];
  st3(gb + 0, lds_get(grd, 0, tid));
  st3(gb + 3, lds_get(grd, 1, tid));
  st3(gb + 6, lds_get(grd, 2, tid));
  st3(gb + 9, gext);
}

// ------------------------------------------------------------------------------------------------
// backbone chain, adjoint.  grid = B, block = 64.  LDS per residue: 3 float4 trig | 3 float4 coords | 3 float4 adj
__global__ __launch_bounds__(PT_WAVE) void nerf_backbone_bwd_kernel(const float *__restrict__ ang,
                                                                    const int64_t *__restrict__ seq,
                                                                    const float *__restrict__ crd,
                                                                    const float *__restrict__ gbb, int L,
                                                                    float *__restrict__ dang) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.x, lane = threadIdx.x;
  ang += (size_t)b * L * 12;
  seq += (size_t)b * L;
  crd += (size_t)b * L * 42;
  gbb += (size_t)b * L * 12;
  dang += (size_t)b * L * 12;
  int32_t dummy = 0;
  const int len = protein_len(seq, L, lane, &dummy);
  if (len < 2) return;
  float *trig = lds, *xyz = lds + (size_t)len * 12, *adj = lds + (size_t)len * 24;
  for (int t = lane; t < len * 6; t += PT_WAVE) {
    int i = t / 6, k = t - i * 6;
    float s, c;
    sincosf(ang[i * 12 + k], &s, &c);
    trig[i * 12 + 2 * k] = s;
    trig[i * 12 + 2 * k + 1] = c;
  }
  for (int t = lane; t < len * 9; t += PT_WAVE) {
    int i = t / 9, k = t - i * 9;
    xyz[i * 12 + k] = crd[i * 42 + k];
  }
  // d/dpsi_i from O_i, written by the side-chain kernel: staged so the serial walk never waits on HBM
  for (int t = lane; t < len; t += PT_WAVE) xyz[t * 12 + 9] = dang[t * 12 + 1];
  for (int t = lane; t < len * 12; t += PT_WAVE) adj[t] = gbb[t];
  __syncthreads();
  if (lane != 0) return;

  auto V = [](const float *p) { return V3{p[0], p[1], p[2]}; };
  const float *G = adj;
  // window: cur = residue i, prv = residue i-1
  int i = len - 1;
  V3 cN = V(G + i * 12), cCA = V(G + i * 12 + 3), cC = V(G + i * 12 + 6);
  for (; i >= 1; --i) {
    V3 pN = V(G + (i - 1) * 12), pCA = V(G + (i - 1) * 12 + 3), pC = V(G + (i - 1) * 12 + 6);
    pC = pC + V(G + i * 12 + 9);                 // CB of residue i hangs off C_{i-1}
    if (i == 1) cN = cN + V(G + 0 * 12 + 9);     // CB of residue 0 hangs off N_1
    const float *tq = trig + i * 12, *tp = trig + (i - 1) * 12;
    V3 xN = V(xyz + i * 12), xCA = V(xyz + i * 12 + 3);
    V3 yN = V(xyz + (i - 1) * 12), yCA = V(xyz + (i - 1) * 12 + 3), yC = V(xyz + (i - 1) * 12 + 6);
    float gth, gchi;
    // C_i = nerf(C_{i-1}, N_i, CA_i; theta = ang[i][3], chi = ang[i][0])
    gth = gchi = 0.f;
    place_bwd(yC, xN, xCA, BL_CA_C, tq[6], tq[7], tq[0], tq[1], cC, pC, cN, cCA, gth, gchi);
    dang[i * 12 + 3] = gth;
    dang[i * 12 + 0] = gchi;
    // CA_i = nerf(CA_{i-1}, C_{i-1}, N_i; theta = ang[i-1][5], chi = ang[i-1][2])
    gth = gchi = 0.f;
    place_bwd(yCA, yC, xN, BL_N_CA, tp[10], tp[11], tp[4], tp[5], cCA, pCA, pC, cN, gth, gchi);
    dang[(i - 1) * 12 + 5] = gth;
    dang[(i - 1) * 12 + 2] = gchi;
    // N_i = nerf(N_{i-1}, CA_{i-1}, C_{i-1}; theta = ang[i-1][4], chi = ang[i-1][1])
    gth = gchi = 0.f;
    place_bwd(yN, yCA, yC, BL_C_N, tp[8], tp[9], tp[2], tp[3], cN, pN, pCA, pC, gth, gchi);
    dang[(i - 1) * 12 + 4] = gth;
    dang[(i - 1) * 12 + 1] = xyz[(i - 1) * 12 + 9] + gchi;  // psi_{i-1} also placed O_{i-1}
    cN = pN;
    cCA = pCA;
    cC = pC;
  }
  // residue 0: N, CA are constants and C is detached from the graph (StructureBuilder.py:185-187):
  // the adjoints left in (cN, cCA, cC) are dropped, ang[0][3] and ang[0][0] get no gradient.
}

}  // namespace

// ================================================================================================
extern "C" {

int ptamd_sidechain_atoms(int residue) { return (residue < 0 || residue > 19) ? -1 : h_pt_nsc[residue]; }

int ptamd_angles_fwd(const float *sincos, float *ang, int64_t n, void *stream) {
  if (n < 0) return PTAMD_ERR_BAD_SHAPE;
  if (n == 0) return PTAMD_OK;
  hipLaunchKernelGGL(angles_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float2 *>(sincos), ang, n);
  return pt_check_launch();
}

int ptamd_angles_bwd(const float *sincos, const float *dang, float *dsincos, int64_t n, void *stream) {
  if (n < 0) return PTAMD_ERR_BAD_SHAPE;
  if (n == 0) return PTAMD_OK;
  hipLaunchKernelGGL(angles_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float2 *>(sincos), dang, reinterpret_cast<float2 *>(dsincos), n);
  return pt_check_launch();
}

size_t ptamd_nerf_workspace_bytes(int B, int L) { return (size_t)(B > 0 ? B : 0) * (L > 0 ? L : 0) * 12 * sizeof(float); }

int ptamd_nerf_fwd(const float *ang, const int64_t *seq, int B, int L, float *crd, int32_t *status, void *stream) {
  if (B <= 0 || L <= 0) return PTAMD_ERR_BAD_SHAPE;
  if (L > MAX_L_CHAIN) return PTAMD_ERR_TOO_LONG;
  size_t lds = (size_t)L * 12 * sizeof(float);
  if (lds > 48 * 1024) {
    PT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(nerf_backbone_fwd_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipLaunchKernelGGL(nerf_backbone_fwd_kernel, dim3(B), dim3(PT_WAVE), lds, (hipStream_t)stream, ang, seq, L, crd,
                     status);
  int rc = pt_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(nerf_sidechain_fwd_kernel, dim3((L + SC_BLOCK - 1) / SC_BLOCK, B), dim3(SC_BLOCK), 0,
                     (hipStream_t)stream, ang, seq, L, crd);
  return pt_check_launch();
}

int ptamd_nerf_bwd(const float *ang, const int64_t *seq, const float *crd, const float *dcrd, int B, int L,
                   float *dang, void *workspace, size_t workspace_bytes, void *stream) {
  if (B <= 0 || L <= 0) return PTAMD_ERR_BAD_SHAPE;
  if (L > MAX_L_CHAIN) return PTAMD_ERR_TOO_LONG;
  if (!workspace || workspace_bytes < ptamd_nerf_workspace_bytes(B, L)) return PTAMD_ERR_WORKSPACE;
  float *gbb = static_cast<float *>(workspace);
  hipLaunchKernelGGL(nerf_sidechain_bwd_kernel, dim3((L + SC_BLOCK - 1) / SC_BLOCK, B), dim3(SC_BLOCK), 0,
                     (hipStream_t)stream, ang, seq, crd, dcrd, L, dang, gbb);
  int rc = pt_check_launch();
  if (rc) return rc;
  size_t lds = (size_t)L * 36 * sizeof(float);
  if (lds > 48 * 1024) {
    PT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(nerf_backbone_bwd_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipLaunchKernelGGL(nerf_backbone_bwd_kernel, dim3(B), dim3(PT_WAVE), lds, (hipStream_t)stream, ang, seq, crd, gbb,
                     L, dang);
  return pt_check_launch();
}

}  // extern "C"
