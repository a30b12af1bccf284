"""Shader-cycle stamps of the phases of a stage in the f16x2 staging GEMM (a TRACE build of the library: temporary
instrumentation, not in the tree - see profiles/r03/r03_gemm_stage_trace.txt).  python profiles/tools/r03_gemm_trace.py {nn|dw} M N K"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K   # noqa: E402
from protein_transformer_amd._lib import lib       # noqa: E402

dev = torch.device("cuda:0")
kind = sys.argv[1]
M, N, Kd = (int(v) for v in sys.argv[2:5])
s12 = (127 + 12) << 23
C = torch.zeros(M, N, device=dev)
if kind == "nn":
    A, B = torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev)
    sa = torch.full((M,), s12, dtype=torch.int32, device=dev)
    sb = torch.full((N,), s12, dtype=torch.int32, device=dev)
    run = lambda: K.gemm(A, B, C, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, arith=K.GEMM_F16X2, a_scale=sa, b_scale=sb)   # noqa: E731
else:
    A, B = torch.randn(Kd, M, device=dev), torch.randn(Kd, N, device=dev)
    u = torch.full((4,), s12, dtype=torch.int32, device=dev)
    run = lambda: K.gemm(A, B, C, M=M, N=N, K=Kd, lda=M, ldb=N, ldc=N, a_kmajor=True, b_kmajor=True, flags=K.EPI_ACCUM,   # noqa: E731
                         split_k=K.pick_split_k(M, N, Kd), arith=K.GEMM_F16X2, a_scale=u, a_scale_stride=0, b_scale=u, b_scale_stride=0)
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
print(f"{kind} {M} x {N} x {Kd}: traced launch {e0.elapsed_time(e1) * 1e3:.1f} us")
t = trace.cpu().numpy().astype(np.int64).reshape(2, 8, 64, 4)
for wg in range(2):
    print(f"--- workgroup {'0' if wg == 0 else '100'}  (shader cycles, s_memtime, low 32 bits; means over stages 8..55)")
    for w in range(8):
        x = t[wg, w]
        d = lambda a, b: ((x[8:56, a] - x[8:56, b]) & 0xFFFFFFFF)   # noqa: E731
        stage = ((x[9:57, 0] - x[8:56, 0]) & 0xFFFFFFFF)
        if w >= 4:   # producer: 0 = top, 1 = loads of the set arrived, 2 = converted and stored to LDS (issue), 3 = refill issued
            print(f"producer {w}: stage {stage.mean():7.0f} [{stage.min():5d}..{stage.max():5d}] | (stamp) {d(1, 0).mean():6.0f}  convert+store+refill {d(2, 1).mean():6.0f}"
                  f"  tail {d(3, 2).mean():6.0f}  barrier wait {(stage - d(3, 0)).mean():6.0f}")
        else:        # consumer: 0 = stage start, 1 = MFMAs issued, 2 = barrier passed
            print(f"consumer {w}: stage {stage.mean():7.0f} [{stage.min():5d}..{stage.max():5d}] | reads+MFMA issue {d(1, 0).mean():6.0f}  barrier wait {d(2, 1).mean():6.0f}"
                  f"  rest {(stage - d(2, 0)).mean():6.0f}")
    np.save(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", f"trace_{kind}_wg{wg}.npy"), t[wg])
