"""One f16x2 GEMM shape with caller scales, looped - the target of the PMC passes of r02_gemm_pmc.sh.
python profiles/tools/r02_gemm_one.py M N K [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K   # noqa: E402

dev = torch.device("cuda:0")
M, N, Kd = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
A = torch.randn(M, Kd, device=dev)
B = torch.randn(N, Kd, device=dev)
C = torch.empty(M, N, device=dev)
sa = torch.full((M,), (127 + 12) << 23, dtype=torch.int32, device=dev)     # 2^12: |x| < 8 -> below 2^15
sb = torch.full((N,), (127 + 12) << 23, dtype=torch.int32, device=dev)


def run():
    K.gemm(A, B, C, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, arith=K.GEMM_F16X2, a_scale=sa, b_scale=sb)


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
print(f"{M} x {N} x {Kd}: {t:.1f} us  {2.0 * M * N * Kd / t / 1e6:.1f} TF/s f32-equivalent")
