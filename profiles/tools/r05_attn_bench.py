"""The attention kernels of the headline step alone, as the step runs them (32 proteins x 512, 8 heads of 64, dropout 0.1,
AUTO = f16x2): K / V as planes written by the QKV product, the forward kernel exporting its dropout decisions, the one-sweep
backward kernel reading both.  PTAMD_LIB_TAG selects an ablation build.  python profiles/tools/r05_attn_bench.py [B L H]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K  # noqa: E402

B, L, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 512, 8)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
D, T = 64 * H, B * L
x = (torch.randn(T, D, generator=g) * torch.exp(0.25 * torch.randn(T, 1, generator=g))).to(dev)
w = (torch.randn(3 * D, D, generator=g) / np.sqrt(D) * 1.1).to(dev)
kv = K.attention_kv_buffers(T, H, dev)
qkv = K.gemm_hp(K.hp_split(x), K.hp_split(w), torch.zeros(T, 3 * D, device=dev), kv=kv, kv_col0=D, kv_heads=H)
seq = torch.randint(0, 20, (B, L), generator=g).to(dev)
seq[1, 400:] = 20
dout = torch.randn(T, D, generator=g).to(dev)
bits = K.attention_keep_bits(B, L, H, dev)


def timed(f, n=50):
    for _ in range(5):
        r = f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        r = f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, r


tf, (out, lse) = timed(lambda: K.attention_fwd(qkv, seq, H, 0.1, 1234, 7, arith=K.GEMM_AUTO, keep_bits=bits, kv=kv))
rs = torch.full((T,), 0x7F000000, dtype=torch.int32, device=dev)
rm = torch.full((4,), 0x7F000000, dtype=torch.int32, device=dev)
tb, dqkv = timed(lambda: K.attention_bwd(qkv, seq, out, dout, lse, H, 0.1, 1234, 7, arith=K.GEMM_AUTO, keep_bits=bits, kv=kv,
                                         row_scale=rs, row_scale_min=rm))
print(f"lib {os.environ.get('PTAMD_LIB_TAG', 'product'):8s} fwd {tf:6.1f} us  bwd (delta + sweep) {tb:6.1f} us   checksums "
      f"{float(out.double().abs().sum()):.6f} {float(dqkv.double().abs().sum()):.6f}")
