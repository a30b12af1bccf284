"""s_memtime stamps per stage of ptamd_gemm_hp (a TRACE build, temporary instrumentation: profiles/r03/r03_gemm_stage_trace.txt).
python profiles/tools/r03_hp_trace.py M N K"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K   # noqa: E402
from protein_transformer_amd._lib import lib       # noqa: E402

dev = torch.device("cuda:0")
M, N, Kd = (int(v) for v in sys.argv[1:4])
a, w = torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev) * 0.05
C = torch.empty(M, N, device=dev)
A, B = K.hp_split(a), K.hp_split(w)
run = lambda: K.gemm_hp(A, B, C)   # noqa: E731
for _ in range(5):
    run()
torch.cuda.synchronize()
trace = torch.zeros(2 * 8 * 64 * 4, dtype=torch.int32, device=dev)
lib().ptamd_debug_trace.argtypes = [ctypes.c_void_p]
lib().ptamd_debug_trace(ctypes.c_void_p(trace.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
lib().ptamd_debug_trace(ctypes.c_void_p(0))
print(f"hp {M} x {N} x {Kd}: traced launch {e0.elapsed_time(e1) * 1e3:.1f} us")
t = trace.cpu().numpy().astype(np.int64).reshape(2, 8, 64, 4)
for wg in range(2):
    print(f"--- workgroup {'0' if wg == 0 else '100'}  (shader cycles; means over stages 4..59, item ends included)")
    for w_ in range(8):
        x = t[wg, w_]
        d = lambda a_, b_: ((x[4:60, a_] - x[4:60, b_]) & 0xFFFFFFFF)   # noqa: E731
        stage = ((x[5:61, 0] - x[4:60, 0]) & 0xFFFFFFFF)
        print(f"wave {w_}: stage {np.median(stage):7.0f} (median; mean {stage.mean():7.0f}) | wait for its DMA {d(1, 0).mean():6.0f}  barrier {d(2, 1).mean():6.0f}"
              f"  reads + MFMAs + DMA issue {d(3, 2).mean():6.0f}  rest (incl. epilogues) {(stage - d(3, 0)).mean():6.0f}")
