"""Staggered start of the two workgroups that share a CU (128 x 128 tiles of gemm_hp3_kernel<EPI, 2>): does one
workgroup's tile epilogue (an HBM burst when every workgroup of the launch reaches it at the same time) hide beside the
other's main loop?  PTAMD_HP_STAGGER_US = delay of the late workgroups, PTAMD_HP_STAGGER_MODE = which are late
(1: the second half of the grid, 0: odd index inside the XCD, 2: odd XCDs).
python profiles/tools/r04_hp_stagger.py [reps] [T]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K   # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
D, F = 512, 2048
VARIANTS = [("256", None, None), ("128", None, None)] + [("128", us, m) for m in (1, 0, 2) for us in (3, 6, 10)] + \
           [("256", 5, 2)]


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


def run_variants(call, label):
    cells = []
    for tile, us, mode in VARIANTS:
        os.environ["PTAMD_HP_TILE"] = tile
        if us is None:
            os.environ.pop("PTAMD_HP_STAGGER_US", None)
        else:
            os.environ["PTAMD_HP_STAGGER_US"] = str(us)
            os.environ["PTAMD_HP_STAGGER_MODE"] = str(mode)
        cells.append(f"{timeit(call):6.1f}")
    for k in ("PTAMD_HP_TILE", "PTAMD_HP_STAGGER_US", "PTAMD_HP_STAGGER_MODE"):
        os.environ.pop(k, None)
    print(f"{label:42s} | " + " ".join(cells), flush=True)


g = torch.Generator(device=dev).manual_seed(3)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)          # noqa: E731
print(f"T = {T}; us per call; columns: " + " ".join(f"{t}" + ("" if us is None else f"/{us}us/m{m}") for t, us, m in VARIANTS))
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
    run_variants(lambda: K.gemm_hp(A, B, C, **kw), f"{name:8s} {T}x{N}x{Kd} {sorted(e)}")
dxs = [("dX ff2", F, D, dict(gate=True)), ("dX ff1", D, F, dict()), ("dX wo", D, D, dict()), ("dX qkv", D, 3 * D, dict())]
for name, Nout, Kc, e in dxs:
    dy, w = rn(T, Kc), rn(Kc, Nout) * 0.05
    gate = torch.relu(rn(T, Nout)) if e.get("gate") else None
    C = torch.empty(T, Nout, device=dev)
    kw = dict(residual=gate, ldr=Nout if gate is not None else 0, flags=K.EPI_GATE if gate is not None else 0,
              gate_scale=1.0 / 0.9 if gate is not None else 0.0)
    A, B = K.hp_split(dy), K.hp_split(w, transposed=True)
    run_variants(lambda: K.gemm_hp(A, B, C, **kw), f"{name:8s} {T}x{Nout}x{Kc} {sorted(e)}")
