"""Attention kernels alone at the benchmark shape (32 proteins x 512 residues, 8 heads of 64, dropout 0.1, AUTO = f16x2):
forward, backward (fused sweep).  python profiles/tools/r04_attn_bench.py [B L H]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K
B, L, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 512, 8)
dev = torch.device("cuda:0")
torch.manual_seed(0)
D = 64 * H
qkv = torch.randn(B * L, 3 * D, device=dev) * 0.7
seq = torch.randint(0, 20, (B, L), device=dev)
seq[1, 400:] = 20
dout = torch.randn(B * L, D, device=dev)
def timed(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): r = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, r
tf, (out, lse) = timed(lambda: K.attention_fwd(qkv, seq, H, 0.1, 1234, 7))
rs = torch.full((B * L,), 0x7F000000, dtype=torch.int32, device=dev); rm = torch.full((4,), 0x7F000000, dtype=torch.int32, device=dev)
tb, dqkv = timed(lambda: K.attention_bwd(qkv, seq, out, dout, lse, H, 0.1, 1234, 7, row_scale=rs, row_scale_min=rm))
print(f"lib {os.environ.get('PTAMD_LIB_TAG', 'product')}: fwd {tf:.1f} us  bwd {tb:.1f} us   checksums {float(out.double().abs().sum()):.6f} {float(dqkv.double().abs().sum()):.6f}")
