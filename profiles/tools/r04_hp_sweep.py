"""ptamd_gemm_hp: the two-buffer kernel (PTAMD_HP_STAGES=2) against the three-stage kernel over K and N with a plain store:
slope = main loop per 32-k stage, intercept = per-tile fixed cost.  python profiles/tools/r04_hp_sweep.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
T = 16384
print("lib", os.environ.get("PTAMD_LIB_TAG", "product"))
for N in (512, 1536, 2048):
    for Kd in (512, 1024, 2048):
        a, w = torch.randn(T, Kd, device=dev, generator=g), torch.randn(N, Kd, device=dev, generator=g) * 0.05
        C = torch.empty(T, N, device=dev)
        A, B = K.hp_split(a), K.hp_split(w)
        os.environ["PTAMD_HP_STAGES"] = "2"
        t2 = timeit(lambda: K.gemm_hp(A, B, C))
        os.environ.pop("PTAMD_HP_STAGES")
        t3 = timeit(lambda: K.gemm_hp(A, B, C))
        fl = 2.0 * T * N * Kd
        print(f"N {N:5d} K {Kd:5d}: two-buffer {t2:7.1f} us {fl / t2 / 1e6:6.1f} TF/s | three-stage {t3:7.1f} us {fl / t3 / 1e6:6.1f} TF/s")
