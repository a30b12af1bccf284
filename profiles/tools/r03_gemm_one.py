"""One f16x2 GEMM shape, looped - the target of the PMC passes of r03_gemm_pmc.sh.
python profiles/tools/r03_gemm_one.py {nn|dw} M N K [reps]
  nn: A [M,K], B [N,K], per-row scales (forward / dX products);  dw: A stored [K,M], B stored [K,N], uniform scales,
  accumulate + fused column sums + split-K as the step's weight-gradient products"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K   # noqa: E402

dev = torch.device("cuda:0")
kind = sys.argv[1]
M, N, Kd = (int(v) for v in sys.argv[2:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
s12 = (127 + 12) << 23                                # 2^12: |x| < 8 -> below 2^15
C = torch.zeros(M, N, device=dev)
if kind == "nn":
    A, B = torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev)
    sa = torch.full((M,), s12, dtype=torch.int32, device=dev)
    sb = torch.full((N,), s12, dtype=torch.int32, device=dev)
    run = lambda: K.gemm(A, B, C, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, arith=K.GEMM_F16X2, a_scale=sa, b_scale=sb)   # noqa: E731
else:
    A, B = torch.randn(Kd, M, device=dev), torch.randn(Kd, N, device=dev)
    u = torch.full((4,), s12, dtype=torch.int32, device=dev)
    cs = torch.zeros(M, device=dev)
    sk = K.pick_split_k(M, N, Kd)
    run = lambda: K.gemm(A, B, C, M=M, N=N, K=Kd, lda=M, ldb=N, ldc=N, a_kmajor=True, b_kmajor=True, flags=K.EPI_ACCUM,   # noqa: E731
                         split_k=sk, colsum=cs, arith=K.GEMM_F16X2, a_scale=u, a_scale_stride=0, b_scale=u, b_scale_stride=0)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / reps * 1e3
print(f"{kind} {M} x {N} x {Kd}: {t:.1f} us  {2.0 * M * N * Kd / t / 1e6:.1f} TF/s f32-equivalent")
