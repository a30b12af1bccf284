"""Time per 16-k stage of the f16x2 GEMM kernel against the number of CUs that run it (ptamd_gemm_args.reserved_cus):
is the stage time of the full chip (0.85-1.0 us against 0.5 us for a workgroup alone) a chip-level effect?
python profiles/tools/r02_gemm_occupancy.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K   # noqa: E402

dev = torch.device("cuda:0")
M, N, Kd = 16384, 2048, 2048
A = torch.randn(M, Kd, device=dev)
B = torch.randn(N, Kd, device=dev)
C = torch.empty(M, N, device=dev)
sa = torch.full((M,), (127 + 12) << 23, dtype=torch.int32, device=dev)
sb = torch.full((N,), (127 + 12) << 23, dtype=torch.int32, device=dev)
tiles = (M // 256) * (N // 128)
for busy in (256, 192, 128, 64, 32, 8, 1):
    K.GEMM_RESERVED_CUS = 256 - busy

    def run():
        K.gemm(A, B, C, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, arith=K.GEMM_F16X2, a_scale=sa, b_scale=sb)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5 if busy >= 32 else 1
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e3
    per_wg = -(-tiles // busy)                       # tiles of the busiest workgroup
    stages = per_wg * (Kd // 16)
    print(f"{busy:4d} CUs busy: {t:9.1f} us, {per_wg:4d} tiles per workgroup -> {t / stages:6.3f} us per 16-k stage "
          f"({2.0 * M * N * Kd / t / 1e6 / busy * 256:6.1f} TF/s f32-equivalent scaled to 256 CUs)")
K.GEMM_RESERVED_CUS = 0
