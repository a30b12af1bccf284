"""Weight-gradient products of one encoder layer (T = 16384): ptamd_gemm (uniform-scale f16x2, splits while staging) against
ptamd_gemm_hp_dw (token-major pre-split operands).  python profiles/tools/r03_gemm_dw_bench.py [reps] [T]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K   # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
D, F = 512, 2048


def timeit(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def uni_scale(x):
    bits = x.abs().max().reshape(1).view(torch.int32)
    return (torch.clamp(268 - (bits >> 23), max=254) << 23).to(torch.int32).repeat(4).contiguous()


print(f"T = {T}")
print(f"{'product':10s} {'M':>6} {'N':>5} | {'ptamd_gemm us':>13} {'TF/s':>6} | {'hp_dw us':>9} {'TF/s':>6} | splits, variants of split_k")
tot = [0.0, 0.0]
for name, M, N in (("dW ff2", D, F), ("dW ff1", F, D), ("dW wo", D, D), ("dW qkv", 3 * D, D)):
    dy, x = torch.randn(T, M, device=dev), torch.randn(T, N, device=dev)
    dw, db = torch.zeros(M, N, device=dev), torch.zeros(M, device=dev)
    sy, sx = uni_scale(dy), uni_scale(x)
    t1 = timeit(lambda: K.linear_bwd_weight(dy, x, dw, db, dy_scale=sy, x_scale=sx))
    Y, X = K.hp_split(dy), K.hp_split(x)
    t2 = timeit(lambda: K.gemm_hp_dw(Y, X, dw, db))
    sk = K.pick_split_k_dw(M, N, T)
    var = []
    for s in (max(1, sk // 2), sk * 2):
        var.append((s, timeit(lambda: K.gemm_hp_dw(Y, X, dw, db, split_k=s))))
    fl = 2.0 * T * M * N
    tot[0] += t1
    tot[1] += t2
    print(f"{name:10s} {M:6d} {N:5d} | {t1:13.1f} {fl / t1 / 1e6:6.1f} | {t2:9.1f} {fl / t2 / 1e6:6.1f} | {sk}, " +
          ", ".join(f"{s}: {t:.1f}" for s, t in var))
print(f"per layer: ptamd_gemm {tot[0]:.0f} us, hp_dw {tot[1]:.0f} us")
