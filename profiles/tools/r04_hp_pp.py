"""Ping-pong variant of the LDS-DMA GEMM (gemm_hp3pp_kernel, PTAMD_HP_PP = pieces of a wavefront issued in its MEM phase)
against gemm_hp3_kernel on the eight forward / dX products of an encoder layer (T = 16384, D = 512, F = 2048, the step's
epilogues) and on long-K squares; every variant's output is compared bit for bit with the three-stage kernel's.
python profiles/tools/r04_hp_pp.py [reps] [T]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K   # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
D, F = 512, 2048
VARIANTS = ["hp3", "6", "4", "3"]


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


def run_variants(call, C, fl, label):
    ref = None
    cells = []
    for v in VARIANTS:
        if v == "hp3":
            os.environ.pop("PTAMD_HP_PP", None)
        else:
            os.environ["PTAMD_HP_PP"] = v
        C.fill_(float("nan"))
        call()
        torch.cuda.synchronize()
        out = C.clone()
        if ref is None:
            ref = out
            same = "ref"
        else:
            same = "bits=" if torch.equal(out, ref) else f"DIFF {float((out - ref).abs().max()):.3e}"
        t = timeit(call)
        cells.append(f"{t:7.1f} us {fl / t / 1e6:6.1f} TF/s {same:8s}")
    os.environ.pop("PTAMD_HP_PP", None)
    print(f"{label:44s} | " + " | ".join(cells), flush=True)


g = torch.Generator(device=dev).manual_seed(3)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)          # noqa: E731
print(f"T = {T}; columns: " + ", ".join("gemm_hp3" if v == "hp3" else f"ping-pong PM={v}" for v in VARIANTS))
fwd = [("qkv fwd", 3 * D, D, dict()),
       ("wo fwd", D, D, dict(res=True, drop=True)),
       ("ff1 fwd", F, D, dict(relu=True, drop=True)),
       ("ff2 fwd", D, F, dict(res=True, drop=True))]
for name, N, Kd, e in fwd:
    a, w, bias = rn(T, Kd), rn(N, Kd) * 0.05, rn(N)
    res = rn(T, N) if e.get("res") else None
    C = torch.empty(T, N, device=dev)
    kw = dict(bias=bias, residual=res, ldr=N if res is not None else 0, flags=K.EPI_RELU if e.get("relu") else 0,
              dropout_p=0.1 if e.get("drop") else 0.0, seed=5, stream_id=1)
    A, B = K.hp_split(a), K.hp_split(w)
    run_variants(lambda: K.gemm_hp(A, B, C, **kw), C, 2.0 * T * N * Kd, f"{name:8s} {T}x{N}x{Kd} {sorted(e)}")
dxs = [("dX ff2", F, D, dict(gate=True)), ("dX ff1", D, F, dict()), ("dX wo", D, D, dict()), ("dX qkv", D, 3 * D, dict())]
for name, Nout, Kc, e in dxs:
    dy, w = rn(T, Kc), rn(Kc, Nout) * 0.05
    gate = torch.relu(rn(T, Nout)) if e.get("gate") else None
    C = torch.empty(T, Nout, device=dev)
    kw = dict(residual=gate, ldr=Nout if gate is not None else 0, flags=K.EPI_GATE if gate is not None else 0,
              gate_scale=1.0 / 0.9 if gate is not None else 0.0)
    A, B = K.hp_split(dy), K.hp_split(w, transposed=True)
    run_variants(lambda: K.gemm_hp(A, B, C, **kw), C, 2.0 * T * Nout * Kc, f"{name:8s} {T}x{Nout}x{Kc} {sorted(e)}")
for M, N, Kd in [(16384, 2048, 2048), (8192, 4096, 4096), (1000, 520, 200), (256, 128, 32), (16384, 512, 96)]:
    a, w = rn(M, Kd), rn(N, Kd) * 0.05
    C = torch.empty(M, N, device=dev)
    A, B = K.hp_split(a), K.hp_split(w)
    run_variants(lambda: K.gemm_hp(A, B, C), C, 2.0 * M * N * Kd, f"plain    {M}x{N}x{Kd}")
