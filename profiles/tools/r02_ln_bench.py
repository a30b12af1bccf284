"""Micro-benchmark of the fused LayerNorm-backward + dropout-backward kernel at the benchmark shape, with and without its
scale outputs (row scales, bound scales, and the atomicMin targets).   python profiles/tools/r02_ln_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K   # noqa: E402

dev = torch.device("cuda:0")
T, D = 16384, 512
x, dy, dres = (torch.randn(T, D, device=dev) for _ in range(3))
gamma = torch.rand(D, device=dev) + 0.5
y, mean, rstd = K.layernorm_fwd(x, gamma, torch.zeros(D, device=dev))
dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
i32 = lambda n: torch.full((n,), 0x7F000000, dtype=torch.int32, device=dev)   # noqa: E731
rs, bs, m1, m2 = i32(T), i32(T), i32(4), i32(4)
bf = torch.ones(1, device=dev)


def timeit(fn, reps=50):
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


for name, kw in (("no scale outputs", {}), ("row scales", dict(row_scale=rs)),
                 ("row + bound scales", dict(row_scale=rs, bound_factor=bf, bound_scale=bs)),
                 ("row + bound scales + minima", dict(row_scale=rs, bound_factor=bf, bound_scale=bs, row_scale_min=m1, bound_scale_min=m2))):
    t = timeit(lambda: K.layernorm_bwd_dropout(dy, x, gamma, mean, rstd, dg, db, dres, 0.1, 7, 3, **kw))
    print(f"layernorm_bwd_dropout [{T} x {D}] {name:30s} {t:7.1f} us   ({5 * T * D * 4 / t / 1e6:.2f} TB/s of 5 T D floats)")
t = timeit(lambda: K.layernorm_fwd(x, gamma, torch.zeros(D, device=dev), row_scale=rs))
print(f"layernorm_fwd with row scales {t:7.1f} us")
