"""Micro-benchmark of the fused attention kernels (split-bf16 arithmetic for dk = 64 / 32, and the exact-f32 kernels) at the benchmark shapes.
python profiles/tools/r02_attn_bench.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K   # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for B, L, H, dk, arith in [(32, 512, 8, 64, None), (8, 1500, 8, 64, None), (16, 256, 8, 32, None), (16, 256, 8, 32, K.GEMM_F32),
                           (32, 512, 8, 32, None), (32, 512, 8, 32, K.GEMM_F32)]:
    D = H * dk
    qkv = torch.randn(B * L, 3 * D, device=dev)
    seq = torch.randint(0, 20, (B, L), device=dev)
    dout = torch.randn(B * L, D, device=dev)

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
    out, lse = K.attention_fwd(qkv, seq, H, 0.1, 7, 3, arith=arith)
    t_f = timeit(lambda: K.attention_fwd(qkv, seq, H, 0.1, 7, 3, arith=arith))
    t_b = timeit(lambda: K.attention_bwd(qkv, seq, out, dout, lse, H, 0.1, 7, 3, arith=arith))
    fl = 4.0 * L * L * dk * B * H
    print(f"B={B} L={L} H={H} dk={dk} arith={arith}: fwd {t_f:8.1f} us ({fl / t_f / 1e6:6.1f} TF/s)   bwd {t_b:8.1f} us ({2.5 * fl / t_b / 1e6:6.1f} TF/s)")
