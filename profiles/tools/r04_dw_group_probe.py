"""What would grouping the four weight-gradient products of a layer into one launch buy?  An upper-bound probe: ONE product with the
same number of 256 x 128 output tiles (96) and the same contraction (T = 16384 tokens), at several K-split counts, against
the four products as the step runs them (8 / 8 / 10 / 32 splits, four reduce launches).
python profiles/tools/r04_dw_group_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K   # noqa: E402

dev = torch.device("cuda:0")
T, D, F = 16384, 512, 2048
g = torch.Generator(device=dev).manual_seed(3)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)          # noqa: E731


def timeit(fn, reps=30):
    for _ in range(3):
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


tot = 0.0
for name, N, Kd in [("dW ff2", D, F), ("dW ff1", F, D), ("dW wo", D, D), ("dW qkv", 3 * D, D)]:
    dy, x = rn(T, N), rn(T, Kd)
    dw, db = torch.zeros(N, Kd, device=dev), torch.zeros(N, device=dev)
    sy, sx = uni_scale(dy), uni_scale(x)
    t = timeit(lambda: K.linear_bwd_weight(dy, x, dw, db, dy_scale=sy, x_scale=sx))
    tot += t
    print(f"{name}: {t:.1f} us (split {K.pick_split_k(N, Kd, T)})")
print(f"the four as the step runs them: {tot:.1f} us, {2.0 * T * (4 * D * F // 2 + 4 * D * D + 0) / 1e6:.0f} MF")
N, Kd = F, 3 * D
dy, x = rn(T, N), rn(T, Kd)
dw, db = torch.zeros(N, Kd, device=dev), torch.zeros(N, device=dev)
sy, sx = uni_scale(dy), uni_scale(x)
fl = 2.0 * T * N * Kd
for sk in (1, 2, 3, 4, 8):
    t = timeit(lambda: K.gemm(dy, x, dw, M=N, N=Kd, K=T, lda=N, ldb=Kd, ldc=Kd, a_kmajor=True, b_kmajor=True, flags=K.EPI_ACCUM,
                              split_k=sk, colsum=db, arith=K.GEMM_F16X2, a_scale=sy, a_scale_stride=0, b_scale=sx, b_scale_stride=0))
    print(f"one product [2048 x 1536], 96 tiles, split {sk}: {96 * sk} items: {t:.1f} us = {fl / t / 1e6:.0f} TF/s f32-eq.")
