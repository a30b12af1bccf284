"""ptamd_gemm_hp (NN, pre-split K-contiguous operands) on the SHAPES of the weight-gradient products (K = 16384 tokens, split-K):
what a dW product would cost if both operands came as transposed planes.  python profiles/tools/r03_hp_longk.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K
dev = torch.device("cuda:0")
def timeit(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
T = 16384
for name, M, N in (("dW ff2", 512, 2048), ("dW ff1", 2048, 512), ("dW wo", 512, 512), ("dW qkv", 1536, 512)):
    a, b = torch.randn(M, T, device=dev), torch.randn(N, T, device=dev)
    A, B = K.hp_split(a), K.hp_split(b)
    C = torch.zeros(M, N, device=dev)
    tiles = ((M + 255) // 256) * ((N + 127) // 128)
    for sk in (max(1, 256 // tiles), max(1, 128 // tiles)):
        t = timeit(lambda: K.gemm_hp(A, B, C, split_k=sk, flags=K.EPI_ACCUM))
        print(f"{name} as NN hp product [{M} x {N} x {T}] split_k {sk}: {t:.1f} us  {2.0 * M * N * T / t / 1e6:.1f} TF/s")
