"""Attention kernels alone at the benchmark shape (32 x 512, 8 heads of 64, dropout 0.1, f16x2): forward with / without
the export of its dropout decisions, fused backward drawing the generator / reading the exported words.
python profiles/tools/r04_attn_bits_bench.py [B L H]"""
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
bits = K.attention_keep_bits(B, L, H, dev)
def timed(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): r = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, r
rs = torch.full((B * L,), 0x7F000000, dtype=torch.int32, device=dev); rm = torch.full((4,), 0x7F000000, dtype=torch.int32, device=dev)
for rep in range(3):
    tf0, (out, lse) = timed(lambda: K.attention_fwd(qkv, seq, H, 0.1, 1234, 7))
    tf1, (out1, lse1) = timed(lambda: K.attention_fwd(qkv, seq, H, 0.1, 1234, 7, keep_bits=bits))
    tb0, d0 = timed(lambda: K.attention_bwd(qkv, seq, out, dout, lse, H, 0.1, 1234, 7, row_scale=rs, row_scale_min=rm))
    tb1, d1 = timed(lambda: K.attention_bwd(qkv, seq, out, dout, lse, H, 0.1, 1234, 7, row_scale=rs, row_scale_min=rm, keep_bits=bits))
    print(f"fwd {tf0:.1f} us, fwd + export {tf1:.1f} us; bwd (generator) {tb0:.1f} us, bwd (stored decisions) {tb1:.1f} us; "
          f"same bits: fwd {torch.equal(out, out1)} bwd {torch.equal(d0, d1)}", flush=True)
