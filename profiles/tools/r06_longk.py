"""Round 6: the long-K activation x weight products of a small batch (FFN-2 forward K = 2048, dX of FFN-1 K = 2048, dX of QKV
K = 1536; N = 512) by K split and tile height - the products that cost twice their flops' share of a 4-protein step.
python profiles/tools/r06_longk.py [T ...]      (PTAMD_LIB_TAG=ti1 selects the build with 64-row tiles)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K   # noqa: E402

dev = torch.device("cuda:0")
Ts = [int(x) for x in sys.argv[1:]] or [2048, 4096]
D, F = 512, 2048


def timeit(fn, reps=30):
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


def row_scale(x, dim=1):
    bits = x.abs().amax(dim).contiguous().view(torch.int32)
    return (torch.clamp(268 - (bits >> 23), max=254) << 23).to(torch.int32).contiguous()


g = torch.Generator(device=dev).manual_seed(3)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)          # noqa: E731
print("lib tag:", os.environ.get("PTAMD_LIB_TAG", "(product)"))
for T in Ts:
    print(f"T = {T}")
    # forward FFN-2: x[T, F] w[D, F]^T + b, dropout + residual
    a, w, bias, res = rn(T, F), rn(D, F) * 0.05, rn(D), rn(T, D)
    C = torch.empty(T, D, device=dev)
    sa, sb = row_scale(a), row_scale(w)
    for sk in (1, 2, 4, 8):
        t = timeit(lambda: K.gemm(a, w, C, M=T, N=D, K=F, lda=F, ldb=F, ldc=D, arith=K.GEMM_F16X2, a_scale=sa, b_scale=sb, bias=bias,
                                  residual=res, ldr=D, dropout_p=0.1, seed=5, stream_id=1, split_k=sk))
        print(f"  ff2 fwd  [{T} x {D} x {F}] drop+res  split {sk}: {t:6.1f} us  {2.0 * T * D * F / t / 1e6:6.1f} TF/s")
    # dX of FFN-1: dz1[T, F] W1[F, D] (row-contiguous B)
    dy, w1 = rn(T, F), rn(F, D) * 0.05
    sa, sb = row_scale(dy), row_scale(w1, dim=0)
    for sk in (1, 2, 4, 8):
        t = timeit(lambda: K.gemm(dy, w1, C, M=T, N=D, K=F, lda=F, ldb=D, ldc=D, b_kmajor=True, arith=K.GEMM_F16X2, a_scale=sa,
                                  b_scale=sb, split_k=sk))
        print(f"  dX ff1   [{T} x {D} x {F}] plain     split {sk}: {t:6.1f} us  {2.0 * T * D * F / t / 1e6:6.1f} TF/s")
    dq, wq = rn(T, 3 * D), rn(3 * D, D) * 0.05
    sa, sb = row_scale(dq), row_scale(wq, dim=0)
    for sk in (1, 2, 3, 6):
        t = timeit(lambda: K.gemm(dq, wq, C, M=T, N=D, K=3 * D, lda=3 * D, ldb=D, ldc=D, b_kmajor=True, arith=K.GEMM_F16X2, a_scale=sa,
                                  b_scale=sb, split_k=sk))
        print(f"  dX qkv   [{T} x {D} x {3 * D}] plain     split {sk}: {t:6.1f} us  {2.0 * T * D * 3 * D / t / 1e6:6.1f} TF/s")
    # for scale: the short-K products of the same layer
    x, wo = rn(T, D), rn(D, D) * 0.05
    sa, sb = row_scale(x), row_scale(wo)
    for sk in (1, 2, 4):
        t = timeit(lambda: K.gemm(x, wo, C, M=T, N=D, K=D, lda=D, ldb=D, ldc=D, arith=K.GEMM_F16X2, a_scale=sa, b_scale=sb, bias=bias,
                                  residual=res, ldr=D, dropout_p=0.1, seed=5, stream_id=1, split_k=sk))
        print(f"  wo fwd   [{T} x {D} x {D}] drop+res  split {sk}: {t:6.1f} us  {2.0 * T * D * D / t / 1e6:6.1f} TF/s")
