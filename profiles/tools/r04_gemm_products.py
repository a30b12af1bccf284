"""Per-product micro-benchmark of the 12 GEMMs of one encoder layer of the benchmark step (T = 16384 tokens, D = 512,
F = 2048) WITH the epilogues and caller-provided scales the step uses: ptamd_gemm (f16x2, splits while staging) against
ptamd_gemm_hp (pre-split operands) where the latter applies - round 4: the two-buffer LDS-DMA kernel of round 3
(PTAMD_HP_STAGES=2) and the three-stage kernel with counted waits (the default).
python profiles/tools/r04_gemm_products.py [reps] [T]"""
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


def row_scale(x, dim=1):
    bits = x.abs().amax(dim).contiguous().view(torch.int32)
    return (torch.clamp(268 - (bits >> 23), max=254) << 23).to(torch.int32).contiguous()


def uni_scale(x):
    bits = x.abs().max().reshape(1).view(torch.int32)
    s = (torch.clamp(268 - (bits >> 23), max=254) << 23).to(torch.int32)
    return s.repeat(4).contiguous()


g = torch.Generator(device=dev).manual_seed(3)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)          # noqa: E731
tot = {"gemm": 0.0, "hp": 0.0}
print(f"T = {T}")
print(f"{'product':10s} {'M':>6} {'N':>5} {'K':>6} {'epilogue':24s} | {'ptamd_gemm us':>13} {'TF/s':>6} | {'hp 2-buf us':>10} {'TF/s':>6} | {'hp3 256x128':>10} {'TF/s':>6} | {'hp3 128x128 x2':>10} {'TF/s':>6}")

fwd = [("qkv fwd", 3 * D, D, dict()),
       ("wo fwd", D, D, dict(res=True, drop=True)),
       ("ff1 fwd", F, D, dict(relu=True, drop=True)),
       ("ff2 fwd", D, F, dict(res=True, drop=True))]
for name, N, Kd, e in fwd:
    a, w, bias = rn(T, Kd), rn(N, Kd) * 0.05, rn(N)
    res = rn(T, N) if e.get("res") else None
    C = torch.empty(T, N, device=dev)
    sa, sb = row_scale(a), row_scale(w)
    kw = dict(bias=bias, residual=res, ldr=N if res is not None else 0, flags=K.EPI_RELU if e.get("relu") else 0,
              dropout_p=0.1 if e.get("drop") else 0.0, seed=5, stream_id=1)
    t1 = timeit(lambda: K.gemm(a, w, C, M=T, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, arith=K.GEMM_F16X2, a_scale=sa, b_scale=sb, **kw))
    A, B = K.hp_split(a), K.hp_split(w)
    os.environ["PTAMD_HP_STAGES"] = "2"
    t2o = timeit(lambda: K.gemm_hp(A, B, C, **kw))
    os.environ.pop("PTAMD_HP_STAGES")
    os.environ["PTAMD_HP_TILE"] = "256"
    t2 = timeit(lambda: K.gemm_hp(A, B, C, **kw))
    os.environ["PTAMD_HP_TILE"] = "128"
    t3 = timeit(lambda: K.gemm_hp(A, B, C, **kw))
    os.environ.pop("PTAMD_HP_TILE")
    fl = 2.0 * T * N * Kd
    tot["gemm"] += t1
    tot["hp"] += t2
    print(f"{name:10s} {T:6d} {N:5d} {Kd:6d} {str(sorted(e)):24s} | {t1:13.1f} {fl / t1 / 1e6:6.1f} | {t2o:10.1f} {fl / t2o / 1e6:6.1f} | {t2:10.1f} {fl / t2 / 1e6:6.1f} | {t3:10.1f} {fl / t3 / 1e6:6.1f}")

# dX = dy[T, N] w[N, K]: output columns K, contraction N
dxs = [("dX ff2", F, D, dict(gate=True)), ("dX ff1", D, F, dict()), ("dX wo", D, D, dict()), ("dX qkv", D, 3 * D, dict())]
for name, Nout, Kc, e in dxs:
    dy, w = rn(T, Kc), rn(Kc, Nout) * 0.05                     # w [contraction, out] = row-contiguous B (b_kmajor)
    gate = torch.relu(rn(T, Nout)) if e.get("gate") else None
    C = torch.empty(T, Nout, device=dev)
    sa, sb = row_scale(dy), row_scale(w, dim=0)
    kw = dict(residual=gate, ldr=Nout if gate is not None else 0, flags=K.EPI_GATE if gate is not None else 0,
              gate_scale=1.0 / 0.9 if gate is not None else 0.0)
    t1 = timeit(lambda: K.gemm(dy, w, C, M=T, N=Nout, K=Kc, lda=Kc, ldb=Nout, ldc=Nout, b_kmajor=True, arith=K.GEMM_F16X2,
                               a_scale=sa, b_scale=sb, **kw))
    A, B = K.hp_split(dy), K.hp_split(w, transposed=True)
    os.environ["PTAMD_HP_STAGES"] = "2"
    t2o = timeit(lambda: K.gemm_hp(A, B, C, **kw))
    os.environ.pop("PTAMD_HP_STAGES")
    os.environ["PTAMD_HP_TILE"] = "256"
    t2 = timeit(lambda: K.gemm_hp(A, B, C, **kw))
    os.environ["PTAMD_HP_TILE"] = "128"
    t3 = timeit(lambda: K.gemm_hp(A, B, C, **kw))
    os.environ.pop("PTAMD_HP_TILE")
    fl = 2.0 * T * Nout * Kc
    tot["gemm"] += t1
    tot["hp"] += t2
    print(f"{name:10s} {T:6d} {Nout:5d} {Kc:6d} {str(sorted(e)):24s} | {t1:13.1f} {fl / t1 / 1e6:6.1f} | {t2o:10.1f} {fl / t2o / 1e6:6.1f} | {t2:10.1f} {fl / t2 / 1e6:6.1f} | {t3:10.1f} {fl / t3 / 1e6:6.1f}")

# dW[N, K] += dy[T, N]^T x[T, K]
dws = [("dW ff2", D, F), ("dW ff1", F, D), ("dW wo", D, D), ("dW qkv", 3 * D, D)]
tdw = 0.0
for name, N, Kd in dws:
    dy, x = rn(T, N), rn(T, Kd)
    dw, db = torch.zeros(N, Kd, device=dev), torch.zeros(N, device=dev)
    sy, sx = uni_scale(dy), uni_scale(x)
    t1 = timeit(lambda: K.linear_bwd_weight(dy, x, dw, db, dy_scale=sy, x_scale=sx))
    fl = 2.0 * T * N * Kd
    tdw += t1
    print(f"{name:10s} {N:6d} {Kd:5d} {T:6d} {'accum + colsum, split-K':24s} | {t1:13.1f} {fl / t1 / 1e6:6.1f} |")
print(f"per layer: fwd + dX  ptamd_gemm {tot['gemm']:.0f} us, gemm_hp {tot['hp']:.0f} us;  dW {tdw:.0f} us")
