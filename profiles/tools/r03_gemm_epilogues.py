"""What the fused epilogues cost: ptamd_gemm (f16x2, caller scales) and ptamd_gemm_hp on three shapes of the step with every
epilogue flavour the step uses.  python profiles/tools/r03_gemm_epilogues.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K   # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T = 16384


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
    return e0.elapsed_time(e1) / reps * 1e3


def row_scale(x):
    bits = x.abs().amax(1).contiguous().view(torch.int32)
    return (torch.clamp(268 - (bits >> 23), max=254) << 23).to(torch.int32).contiguous()


print(f"lib tag: {os.environ.get('PTAMD_LIB_TAG', '(product)')}")
print(f"{'N':>5} {'K':>5} {'epilogue':22s} | {'ptamd_gemm us':>13} | {'gemm_hp us':>10}")
for N, Kd in ((2048, 512), (1536, 512), (512, 512), (512, 2048)):
    a, w, bias = torch.randn(T, Kd, device=dev), torch.randn(N, Kd, device=dev) * 0.05, torch.randn(N, device=dev)
    res = torch.randn(T, N, device=dev)
    C = torch.empty(T, N, device=dev)
    sa, sb = row_scale(a), row_scale(w)
    A, B = K.hp_split(a), K.hp_split(w)
    flavours = [("plain", dict()), ("bias", dict(bias=bias)), ("bias relu", dict(bias=bias, flags=K.EPI_RELU)),
                ("bias relu drop", dict(bias=bias, flags=K.EPI_RELU, dropout_p=0.1, seed=5, stream_id=1)),
                ("gate", dict(residual=res, ldr=N, flags=K.EPI_GATE, gate_scale=1.1)),
                ("bias drop res", dict(bias=bias, residual=res, ldr=N, dropout_p=0.1, seed=5, stream_id=1))]
    for name, kw in flavours:
        t1 = timeit(lambda: K.gemm(a, w, C, M=T, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, arith=K.GEMM_F16X2, a_scale=sa, b_scale=sb, **kw))
        t2 = timeit(lambda: K.gemm_hp(A, B, C, **kw))
        print(f"{N:5d} {Kd:5d} {name:22s} | {t1:13.1f} | {t2:10.1f}")
