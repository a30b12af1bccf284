"""Micro-benchmark: ptamd_gemm_hp (pre-split operands, LDS-DMA) against ptamd_gemm in the f16x2 / bf16x3 arithmetics on
the product shapes of the benchmark step (T = 16384 tokens).  python profiles/tools/r02_gemm_hp_bench.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K   # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T = 16384
shapes = [(T, 1536, 512), (T, 512, 512), (T, 2048, 512), (T, 512, 2048), (T, 512, 1536), (4096, 4096, 4096), (T, 2048, 2048)]


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"{'M':>6} {'N':>5} {'K':>5} | {'hp us':>8} {'TF/s':>7} | {'hp+epi':>8} | {'f16x2 us':>9} {'TF/s':>7} | {'bf16x3 us':>9} {'TF/s':>7} | split A us")
for M, N, Kd in shapes:
    zero = os.environ.get("HP_BENCH_ZERO") == "1"
    a = torch.zeros(M, Kd, device=dev) if zero else torch.randn(M, Kd, device=dev)
    b = torch.zeros(N, Kd, device=dev) if zero else torch.randn(N, Kd, device=dev) * 0.05
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    C = torch.empty(M, N, device=dev)
    A, B = K.hp_split(a), K.hp_split(b)
    fl = 2.0 * M * N * Kd
    t_hp = timeit(lambda: K.gemm_hp(A, B, C))
    t_hpe = timeit(lambda: K.gemm_hp(A, B, C, bias=bias, residual=res, ldr=N, dropout_p=0.1, seed=5, stream_id=1))
    t_f16 = timeit(lambda: K.gemm(a, b, C, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, arith=K.GEMM_F16X2))
    t_bf = timeit(lambda: K.gemm(a, b, C, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, arith=K.GEMM_BF16X3))
    t_sp = timeit(lambda: K.hp_split(a, out=A))
    print(f"{M:6d} {N:5d} {Kd:5d} | {t_hp * 1e3:8.1f} {fl / t_hp / 1e9:7.1f} | {t_hpe * 1e3:8.1f} | {t_f16 * 1e3:9.1f} {fl / t_f16 / 1e9:7.1f} | "
          f"{t_bf * 1e3:9.1f} {fl / t_bf / 1e9:7.1f} | {t_sp * 1e3:8.1f}")
