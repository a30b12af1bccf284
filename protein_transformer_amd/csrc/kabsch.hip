// Batched RMSD after optimal superposition (Kabsch) for gfx950: the evaluation metric `rmsd-full`.
//
// Replaces, for a whole batch on the device and without a host round trip per protein,
//   rmsd(a, b)            /root/reference/protein_transformer/losses.py:281-286
//                         (ProDy: t = calcTransformation(a, b); calcRMSD(t.apply(a), b))
// as it is reached from drmsd_work(return_rmsd=True) (losses.py:94-96) under eval_epoch (train.py:114-135): `a` are the
// predicted atoms whose truth is present, `b` the true ones.
//
// The optimally superposed RMSD needs no rotation matrix: with a, b centred, H = sum_i a_i b_i^T and singular values
// s1 >= s2 >= s3 of H,
//     rmsd^2 = ( sum |a_i|^2 + sum |b_i|^2 - 2 (s1 + s2 + sign(det H) s3) ) / n            (Kabsch 1976 / 1978).
// One workgroup per protein: a single HBM-bound sweep over its atom slots accumulates the 18 moments
// (n, sum a, sum b, sum a b^T, sum |a|^2, sum |b|^2) in fp64 (wave shuffles + one LDS step, fixed order); one lane then
// centres them, diagonalises H^T H with cyclic Jacobi rotations in fp64 and writes the protein's RMSD.
// Algorithmic traffic: 24 B per atom slot (pred + true read once), one float written per protein.
#include "common.h"

namespace {

constexpr int KB = 256;      // threads per protein
constexpr int NMOM = 18;     // n, a(3), b(3), H(9), |a|^2, |b|^2

// eigenvalues of the symmetric 3x3 matrix m (upper triangle used), cyclic Jacobi; fp64, unsorted
__device__ void sym3_eigenvalues(double m[3][3], double ev[3]) {
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = fabs(m[0][1]) + fabs(m[0][2]) + fabs(m[1][2]);
    if (off <= 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = m[p][q];
        if (apq == 0.0) continue;
        const double theta = (m[q][q] - m[p][p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        const int r = 3 - p - q;  // the third index
        const double app = m[p][p], aqq = m[q][q];
        m[p][p] = app - t * apq;
        m[q][q] = aqq + t * apq;
        m[p][q] = m[q][p] = 0.0;
        const double arp = m[r][p], arq = m[r][q];
        m[r][p] = m[p][r] = c * arp - s * arq;
        m[r][q] = m[q][r] = s * arp + c * arq;
      }
  }
  ev[0] = m[0][0];
  ev[1] = m[1][1];
  ev[2] = m[2][2];
}

__global__ __launch_bounds__(KB) void kabsch_rmsd_kernel(const float *__restrict__ pred, const float *__restrict__ truth,
                                                         const int64_t *__restrict__ seq, int L, float *__restrict__ out) {
  __shared__ double s_part[KB / 64][NMOM];
  const int b = blockIdx.x, tid = threadIdx.x;
  const size_t nslot = (size_t)L * PTAMD_NUM_SLOTS;
  pred += (size_t)b * nslot * 3;
  truth += (size_t)b * nslot * 3;
  seq += (size_t)b * L;
  double mom[NMOM];
#pragma unroll
  for (int k = 0; k < NMOM; ++k) mom[k] = 0.0;
  for (size_t s = tid; s < nslot; s += KB) {
    if (seq[s / PTAMD_NUM_SLOTS] == PTAMD_PAD_ID) continue;  // batch padding (collate pads the truth with zeros, not NaN)
    const float tx = truth[s * 3], ty = truth[s * 3 + 1], tz = truth[s * 3 + 2];
    if (isnan(tx) || isnan(ty) || isnan(tz)) continue;        // atom absent from the truth (losses.py:70-72)
    const double a[3] = {(double)pred[s * 3], (double)pred[s * 3 + 1], (double)pred[s * 3 + 2]};
    const double c[3] = {(double)tx, (double)ty, (double)tz};
    mom[0] += 1.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      mom[1 + i] += a[i];
      mom[4 + i] += c[i];
#pragma unroll
      for (int j = 0; j < 3; ++j) mom[7 + 3 * i + j] += a[i] * c[j];
    }
    mom[16] += a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    mom[17] += c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
  }
#pragma unroll
  for (int k = 0; k < NMOM; ++k) {
    const double v = wave_sum_d(mom[k]);
    if ((tid & 63) == 0) s_part[tid >> 6][k] = v;
  }
  __syncthreads();
  if (tid != 0) return;
  double m[NMOM];
  for (int k = 0; k < NMOM; ++k) {
    double v = 0.0;
    for (int w = 0; w < KB / 64; ++w) v += s_part[w][k];
    m[k] = v;
  }
  const double n = m[0];
  if (n < 1.0) {
    out[b] = __builtin_nanf("");
    return;
  }
  double H[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) H[i][j] = m[7 + 3 * i + j] - m[1 + i] * m[4 + j] / n;
  const double e0 = (m[16] - (m[1] * m[1] + m[2] * m[2] + m[3] * m[3]) / n) + (m[17] - (m[4] * m[4] + m[5] * m[5] + m[6] * m[6]) / n);
  double K[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) K[i][j] = H[0][i] * H[0][j] + H[1][i] * H[1][j] + H[2][i] * H[2][j];  // H^T H
  double ev[3];
  sym3_eigenvalues(K, ev);
  double s0 = sqrt(fmax(ev[0], 0.0)), s1 = sqrt(fmax(ev[1], 0.0)), s2 = sqrt(fmax(ev[2], 0.0));
  double smin = fmin(s0, fmin(s1, s2));
  const double det = H[0][0] * (H[1][1] * H[2][2] - H[1][2] * H[2][1]) - H[0][1] * (H[1][0] * H[2][2] - H[1][2] * H[2][0]) +
                     H[0][2] * (H[1][0] * H[2][1] - H[1][1] * H[2][0]);
  const double trace = s0 + s1 + s2 - (det < 0.0 ? 2.0 * smin : 0.0);  // s1 + s2 + sign(det) s3
  out[b] = (float)sqrt(fmax(e0 - 2.0 * trace, 0.0) / n);
}

}  // namespace

extern "C" int ptamd_kabsch_rmsd(const float *pred_crd, const float *true_crd, const int64_t *seq, int B, int L, float *rmsd,
                                 void *stream) {
  if (B <= 0 || L <= 0) return PTAMD_ERR_BAD_SHAPE;
  if (!pred_crd || !true_crd || !seq || !rmsd) return PTAMD_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(kabsch_rmsd_kernel, dim3(B), dim3(KB), 0, (hipStream_t)stream, pred_crd, true_crd, seq, L, rmsd);
  return pt_check_launch();
}
