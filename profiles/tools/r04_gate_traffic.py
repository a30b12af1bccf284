"""Upper bound of what a 1-bit gate would give the gated dX product of FFN layer 2 (dz1 = (dy2 W2) * (f1 > 0) / (1 - p)):
the same product with the gate operand read from ONE row (row stride 0: 8 KB, cache resident) instead of the [T, 2048]
fp32 tensor, same arithmetic per element.  python profiles/tools/r04_gate_traffic.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K
dev = torch.device("cuda:0")
T, D, F = 16384, 512, 2048
g = torch.Generator(device=dev).manual_seed(3)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)          # noqa: E731
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def row_scale(x, dim=1):
    bits = x.abs().amax(dim).contiguous().view(torch.int32)
    return (torch.clamp(268 - (bits >> 23), max=254) << 23).to(torch.int32).contiguous()
dy, w = rn(T, D), rn(D, F) * 0.05
gate = torch.relu(rn(T, F))
C = torch.empty(T, F, device=dev)
sa, sb = row_scale(dy), row_scale(w, dim=0)
kw = dict(M=T, N=F, K=D, lda=D, ldb=F, ldc=F, b_kmajor=True, arith=K.GEMM_F16X2, a_scale=sa, b_scale=sb)
for rep in range(3):
    t_full = timeit(lambda: K.gemm(dy, w, C, residual=gate, ldr=F, flags=K.EPI_GATE, gate_scale=1.0 / 0.9, **kw))
    t_row = timeit(lambda: K.gemm(dy, w, C, residual=gate, ldr=0, flags=K.EPI_GATE, gate_scale=1.0 / 0.9, **kw))
    t_plain = timeit(lambda: K.gemm(dy, w, C, **kw))
    print(f"gated dX (staging kernel): gate [T, F] {t_full:.1f} us, gate from one row {t_row:.1f} us, no gate {t_plain:.1f} us", flush=True)
