// run one GEMM shape in a loop for ~N seconds: ./gloop lib.so mode M N K seconds
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>
#include "../../include/ptamd.h"
int main(int argc, char **argv) {
  void *h = dlopen(argv[1], RTLD_NOW);
  auto gemm = (int (*)(const ptamd_gemm_args *, void *))dlsym(h, "ptamd_gemm");
  auto setm = (int (*)(int))dlsym(h, "ptamd_gemm_set_mode");
  setm(atoi(argv[2]));
  int M = atoi(argv[3]), N = atoi(argv[4]), K = atoi(argv[5]); double secs = atof(argv[6]);
  float *A, *B, *C;
  hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
  std::vector<float> hv((size_t)M * K > (size_t)N * K ? (size_t)M * K : (size_t)N * K);
  for (size_t i = 0; i < hv.size(); ++i) hv[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(A, hv.data(), (size_t)M * K * 4, hipMemcpyHostToDevice); hipMemcpy(B, hv.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
  ptamd_gemm_args a = {};
  a.M = M; a.N = N; a.K = K; a.A = A; a.lda = K; a.B = B; a.ldb = K; a.C = C; a.ldc = N; a.split_k = 1;
  auto wsb = (size_t (*)(int, int, int))dlsym(h, "ptamd_gemm_workspace_bytes");   // (row scales of the f16x2 arithmetic)
  a.workspace_bytes = wsb(M, N, 1);
  hipMalloc(&a.workspace, a.workspace_bytes);
  if (gemm(&a, 0)) { printf("gemm error\n"); return 1; }
  auto t0 = std::chrono::steady_clock::now();
  long n = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    for (int i = 0; i < 50; ++i) gemm(&a, 0);
    hipDeviceSynchronize(); n += 50;
  }
  double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("%s mode %s: %.1f us/gemm, %.1f TF/s\n", argv[1], argv[2], el / n * 1e6, 2.0 * M * N * K / (el / n) / 1e12);
  return 0;
}
