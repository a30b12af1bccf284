// dlopen a libptamd variant and time ptamd_gemm on the training-step shapes: ./gbench lib.so [mode]
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/ptamd.h"
int main(int argc, char **argv) {
  void *h = dlopen(argv[1], RTLD_NOW);
  if (!h) { printf("dlopen failed %s\n", dlerror()); return 1; }
  int mode = argc > 2 ? atoi(argv[2]) : 1;
  const bool with_colsum = getenv("GB_COLSUM") != nullptr;
  float *csum; hipMalloc(&csum, 1 << 20);
  auto gemm = (int (*)(const ptamd_gemm_args *, void *))dlsym(h, "ptamd_gemm");
  auto setm = (int (*)(int))dlsym(h, "ptamd_gemm_set_mode");
  auto wsb = (size_t (*)(int, int, int))dlsym(h, "ptamd_gemm_workspace_bytes");
  setm(mode);
  const int T = 16384;
  struct S { const char *n; int M, N, K, ak, bk, split; } sh[] = {
    {"fwd qkv", T, 1536, 512, 0, 0, 1}, {"fwd wo", T, 512, 512, 0, 0, 1}, {"fwd ff1", T, 2048, 512, 0, 0, 1}, {"fwd ff2", T, 512, 2048, 0, 0, 1},
    {"dX ff1", T, 512, 2048, 0, 1, 1}, {"dX qkv", T, 512, 1536, 0, 1, 1}, {"dX ff2", T, 2048, 512, 0, 1, 1}, {"dX wo", T, 512, 512, 0, 1, 1},
    {"dW ff1", 2048, 512, T, 1, 1, 8}, {"dW qkv", 1536, 512, T, 1, 1, 10}, {"dW wo", 512, 512, T, 1, 1, 32}, {"dW ff2", 512, 2048, T, 1, 1, 8},
    {"sq4096", 4096, 4096, 4096, 0, 0, 1},
    {"tiny1", 256, 128, 16, 0, 0, 1}, {"tile512", 256, 128, 512, 0, 0, 1}, {"full16", T, 512, 16, 0, 0, 1}, {"full64", T, 512, 64, 0, 0, 1},
    {"full128", T, 512, 128, 0, 0, 1}, {"full256", T, 512, 256, 0, 0, 1}, {"full1024", T, 512, 1024, 0, 0, 1}};
  float *A, *B, *C; void *ws;
  size_t big = (size_t)T * 2048;
  hipMalloc(&A, big * 4); hipMalloc(&B, big * 4); hipMalloc(&C, big * 4); hipMalloc(&ws, (size_t)600 << 20);
  std::vector<float> hv(big);
  for (size_t i = 0; i < big; ++i) hv[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(A, hv.data(), big * 4, hipMemcpyHostToDevice); hipMemcpy(B, hv.data(), big * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double tot = 0;
  // step weights: per layer 1x each of fwd qkv, wo, ff1, ff2; dX ff1(=dF1*W1 -> N512 K2048), dX qkv, dX ff2, dX wo; dW x4
  const char *only = argc > 3 ? argv[3] : nullptr;
  for (auto &s : sh) {
    if (only && strcmp(only, s.n)) continue;
    ptamd_gemm_args a = {};
    a.M = s.M; a.N = s.N; a.K = s.K; a.A = A; a.lda = s.ak ? s.M : s.K; a.a_kmajor = s.ak; a.B = B; a.ldb = s.bk ? s.N : s.K; a.b_kmajor = s.bk;
    a.C = C; a.ldc = s.N; a.split_k = s.split; if (with_colsum && s.ak) { const char *m = getenv("GB_COLSUM"); if (m[0] != 'a') a.colsum = csum; if (m[0] != 'c') a.flags = PTAMD_EPI_ACCUM; } a.workspace = ws; a.workspace_bytes = wsb(s.M, s.N, s.split);
    for (int i = 0; i < 3; ++i) if (gemm(&a, 0)) { printf("gemm error\n"); return 1; }
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    const int n = 20;
    for (int i = 0; i < n; ++i) gemm(&a, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double us = ms * 1e3 / n;
    if (s.n[0] == 'f' && s.n[1] == 'w' || s.n[0] == 'd') tot += us;
    printf("  %-8s M%-5d N%-4d K%-5d split %2d: %7.1f us %6.1f TF/s\n", s.n, s.M, s.N, s.K, s.split, us, 2.0 * s.M * s.N * s.K / us / 1e6);
  }
  printf("%s mode %d: sum of the 12 step shapes %.0f us (x6 layers = %.2f ms)\n", argv[1], mode, tot, tot * 6e-3);
  return 0;
}
