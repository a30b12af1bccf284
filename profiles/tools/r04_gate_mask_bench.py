"""The gated dX product of FFN layer 2 (dz1 = (dy2 W2) * (f1 > 0) / (1 - p), T = 16384, 512 -> 2048) gated by the fp32
activation f1 against gated by its 1-bit mask (written by the FFN-1 product's epilogue), on the staging kernel (the step's)
and on the LDS-DMA kernel; and the FFN-1 product with / without writing the mask.  python profiles/tools/r04_gate_mask_bench.py"""
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
h2, w1, b1 = rn(T, D), rn(F, D) * 0.05, rn(F) * 0.1
A, B = K.hp_split(h2), K.hp_split(w1)
f1 = torch.empty(T, F, device=dev)
mask = K.gate_mask_buffer(T, F, dev)
kw = dict(bias=b1, flags=K.EPI_RELU, dropout_p=0.1, seed=5, stream_id=1)
dy, w2 = rn(T, D), rn(D, F) * 0.05
C = torch.empty(T, F, device=dev)
sa, sb = row_scale(dy), row_scale(w2, dim=0)
gk = dict(M=T, N=F, K=D, lda=D, ldb=F, ldc=F, b_kmajor=True, arith=K.GEMM_F16X2, a_scale=sa, b_scale=sb, flags=K.EPI_GATE, gate_scale=1.0 / 0.9)
Ady, Bw2 = K.hp_split(dy), K.hp_split(w2, transposed=True)
for rep in range(3):
    t0 = timeit(lambda: K.gemm_hp(A, B, f1, **kw))
    t1 = timeit(lambda: K.gemm_hp(A, B, f1, gate_mask_out=mask, **kw))
    ta = timeit(lambda: K.gemm(dy, w2, C, residual=f1, ldr=F, **gk))
    ref = C.clone()
    tb = timeit(lambda: K.gemm(dy, w2, C, gate_mask=mask, **gk))
    same = torch.equal(ref, C)
    tc = timeit(lambda: K.gemm_hp(Ady, Bw2, C, residual=f1, ldr=F, flags=K.EPI_GATE, gate_scale=1.0 / 0.9))
    td = timeit(lambda: K.gemm_hp(Ady, Bw2, C, gate_mask=mask, flags=K.EPI_GATE, gate_scale=1.0 / 0.9))
    print(f"FFN-1 {t0:.1f} us, + mask export {t1:.1f} us | gated dX staging: by f1 {ta:.1f} us, by mask {tb:.1f} us (same bits {same}) | "
          f"gated dX LDS-DMA kernel: by f1 {tc:.1f} us, by mask {td:.1f} us", flush=True)
