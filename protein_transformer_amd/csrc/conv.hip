// Length-preserving 1-D sequence convolution of the `conv-enc` front end as im2col + the f32 MFMA GEMM.
//
// Replaces torch.nn.Conv1d(d_in, d_out, k, padding=(k-1)/2) on [B, C, L] tensors
//   /root/reference/protein_transformer/models/convolutional_encoder.py:92-123,125-129
// (a stack of such layers with NO activation in between, applied between the embedding and the encoder layers)
// and the doubled positional add of its one-hot variant (`enc_output += positional_enc(enc_output)`, :118-119).
//
// Activations stay token-major [T = B*L, C] like everywhere else in this library, so
//     y[t, co] = bias[co] + sum_{j < k} sum_{c} W[co, c, j] * x[b, l + j - pad, c]          (zero outside 0 <= l' < L)
// is ONE GEMM  y = col @ W2^T  with col[t, j*C + c] = x[b, l + j - pad, c]  (im2col, 16-byte gathers) and
// W2[co, j*C + c] = W[co, c, j] (weights re-packed every step: they are small).  Backward: dW2 = dy^T col,
// dcol = dy W2, dx = col2im(dcol) as a gather (no atomics).  Channel counts are padded to a multiple of 4.
#include "common.h"

namespace {

// col [T, k*Cp] from x [T, C] (row stride ldx); Cp = C rounded up to 4, padded channels are zero
__global__ void im2col1d_kernel(const float *__restrict__ x, int ldx, int L, int C, int Cp, int k, int64_t n4,
                                float *__restrict__ col) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of col
  if (i >= n4) return;
  const int c4 = Cp >> 2, per_row = k * c4;
  const int64_t t = i / per_row;
  const int r = (int)(i - t * per_row), j = r / c4, c = (r - j * c4) * 4;
  const int l = (int)(t % L), ls = l + j - (k - 1) / 2;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ls >= 0 && ls < L) {
    const float *src = x + (t + (ls - l)) * ldx + c;
    if (c + 3 < C) v = *reinterpret_cast<const float4 *>(src);
    else {
      if (c < C) v.x = src[0];
      if (c + 1 < C) v.y = src[1];
      if (c + 2 < C) v.z = src[2];
    }
  }
  *reinterpret_cast<float4 *>(col + i * 4) = v;
}

// dx [T, C] = sum_j dcol[t - (j - pad), j*Cp + c] over the source rows that lie in the same protein
__global__ void col2im1d_kernel(const float *__restrict__ dcol, int L, int C, int Cp, int k, int64_t n,
                                float *__restrict__ dx, int lddx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one element of dx
  if (i >= n) return;
  const int64_t t = i / C;
  const int c = (int)(i - t * C), l = (int)(t % L), pad = (k - 1) / 2;
  float s = 0.f;
  for (int j = 0; j < k; ++j) {
    const int lo = l - (j - pad);  // output position whose window slot j looked at l
    if (lo >= 0 && lo < L) s += dcol[(t + (lo - l)) * (int64_t)(k * Cp) + j * Cp + c];
  }
  dx[t * lddx + c] = s;
}

// W [Co, C, k] -> W2 [Co, k*Cp]
__global__ void conv_pack_kernel(const float *__restrict__ w, int Co, int C, int Cp, int k, float *__restrict__ w2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Co * k * Cp) return;
  const int co = i / (k * Cp), r = i - co * (k * Cp), j = r / Cp, c = r - j * Cp;
  w2[i] = c < C ? w[((size_t)co * C + c) * k + j] : 0.f;
}
// dW [Co, C, k] += dW2 [Co, k*Cp]
__global__ void conv_unpack_add_kernel(const float *__restrict__ dw2, int Co, int C, int Cp, int k, float *__restrict__ dw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Co * C * k) return;
  const int co = i / (C * k), r = i - co * (C * k), c = r / k, j = r - c * k;
  dw[i] += dw2[(size_t)co * (k * Cp) + j * Cp + c];
}

// one-hot rows: x [T, Cp] = (c == seq[t])
__global__ void onehot_kernel(const int64_t *__restrict__ seq, int64_t T, int Cp, float *__restrict__ x) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * Cp) return;
  const int64_t t = i / Cp;
  x[i] = (int64_t)(i - t * Cp) == seq[t] ? 1.f : 0.f;
}

// y = x + dropout(x + pe[l])   (convolutional_encoder.py:118-119 with Sublayers.py:59-62); 4 channels per thread
constexpr uint32_t STREAM_POSADD = 0xE3u;
__global__ void posenc_add_fwd_kernel(const float *__restrict__ x, const float *__restrict__ pe, int L, int D, int64_t n4,
                                      float p, uint64_t seed, float *__restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int d4 = D >> 2;
  const int64_t t = i / d4;
  const int c = (int)(i - t * d4) * 4, l = (int)(t % L);
  const float4 a = *reinterpret_cast<const float4 *>(x + i * 4), q = *reinterpret_cast<const float4 *>(pe + (size_t)l * D + c);
  float xv[4] = {a.x, a.y, a.z, a.w}, pv[4] = {q.x, q.y, q.z, q.w}, o[4];
  if (p > 0.f) {
    const uint32_t thr = dropout_threshold(p);
    const float ks = 1.f / (1.f - p);
    const uint4 r = pt_rand4(seed, (uint64_t)i, STREAM_POSADD);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = xv[k] + (w[k] >= thr ? (xv[k] + pv[k]) * ks : 0.f);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = xv[k] + (xv[k] + pv[k]);
  }
  *reinterpret_cast<float4 *>(y + i * 4) = make_float4(o[0], o[1], o[2], o[3]);
}
__global__ void posenc_add_bwd_kernel(const float *__restrict__ dy, int64_t n4, float p, uint64_t seed,
                                      float *__restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 g = *reinterpret_cast<const float4 *>(dy + i * 4);
  float gv[4] = {g.x, g.y, g.z, g.w}, o[4];
  if (p > 0.f) {
    const uint32_t thr = dropout_threshold(p);
    const float ks = 1.f / (1.f - p);
    const uint4 r = pt_rand4(seed, (uint64_t)i, STREAM_POSADD);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = gv[k] * (1.f + (w[k] >= thr ? ks : 0.f));
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = 2.f * gv[k];
  }
  *reinterpret_cast<float4 *>(dx + i * 4) = make_float4(o[0], o[1], o[2], o[3]);
}

}  // namespace

extern "C" {

int ptamd_im2col1d(const float *x, int ldx, int B, int L, int C, int k, float *col, void *stream) {
  if (B <= 0 || L <= 0 || C <= 0 || k <= 0 || !(k & 1) || (ldx & 3)) return PTAMD_ERR_BAD_SHAPE;
  const int Cp = (C + 3) & ~3;
  const int64_t n4 = (int64_t)B * L * k * (Cp >> 2);
  hipLaunchKernelGGL(im2col1d_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, L, C,
                     Cp, k, n4, col);
  return pt_check_launch();
}

int ptamd_col2im1d(const float *dcol, int B, int L, int C, int k, float *dx, int lddx, void *stream) {
  if (B <= 0 || L <= 0 || C <= 0 || k <= 0 || !(k & 1)) return PTAMD_ERR_BAD_SHAPE;
  const int Cp = (C + 3) & ~3;
  const int64_t n = (int64_t)B * L * C;
  hipLaunchKernelGGL(col2im1d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dcol, L, C, Cp,
                     k, n, dx, lddx);
  return pt_check_launch();
}

int ptamd_conv_weight_pack(const float *w, int Co, int C, int k, float *w2, void *stream) {
  if (Co <= 0 || C <= 0 || k <= 0) return PTAMD_ERR_BAD_SHAPE;
  const int Cp = (C + 3) & ~3, n = Co * k * Cp;
  hipLaunchKernelGGL(conv_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, Co, C, Cp, k, w2);
  return pt_check_launch();
}

int ptamd_conv_weight_unpack_add(const float *dw2, int Co, int C, int k, float *dw, void *stream) {
  if (Co <= 0 || C <= 0 || k <= 0) return PTAMD_ERR_BAD_SHAPE;
  const int Cp = (C + 3) & ~3, n = Co * C * k;
  hipLaunchKernelGGL(conv_unpack_add_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dw2, Co, C, Cp, k, dw);
  return pt_check_launch();
}

int ptamd_onehot(const int64_t *seq, int64_t T, int C, float *x, void *stream) {
  if (T <= 0 || C <= 0) return PTAMD_ERR_BAD_SHAPE;
  const int Cp = (C + 3) & ~3;
  hipLaunchKernelGGL(onehot_kernel, dim3((unsigned)((T * Cp + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seq, T, Cp, x);
  return pt_check_launch();
}

int ptamd_posenc_add_fwd(const float *x, const float *pe, int B, int L, int D, float dropout_p, uint64_t seed, float *y,
                         void *stream) {
  if (B <= 0 || L <= 0 || D <= 0 || (D & 3)) return PTAMD_ERR_BAD_SHAPE;
  const int64_t n4 = (int64_t)B * L * (D >> 2);
  hipLaunchKernelGGL(posenc_add_fwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, pe, L,
                     D, n4, dropout_p, seed, y);
  return pt_check_launch();
}

int ptamd_posenc_add_bwd(const float *dy, int64_t n, float dropout_p, uint64_t seed, float *dx, void *stream) {
  if (n <= 0 || (n & 3)) return PTAMD_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(posenc_add_bwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy,
                     n / 4, dropout_p, seed, dx);
  return pt_check_launch();
}

}  // extern "C"
