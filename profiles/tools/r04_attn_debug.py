"""Where does the fused attention backward differ from the two-kernel path?  python profiles/tools/r04_attn_debug.py B L H p"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K
dev = torch.device("cuda:0")
B, L, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
p = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
D = 64 * H
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B * L, 3 * D, generator=g).to(dev)
dout = torch.randn(B * L, D, generator=g).to(dev)
seq = torch.randint(0, 20, (B, L), generator=g).to(dev)
o, lse = K.attention_fwd(qkv, seq, H, p, 5, 2, arith=K.GEMM_F16X2)
res = {}
for f in ("0", "1"):
    os.environ["PTAMD_ATTN_FUSED"] = f
    res[f] = K.attention_bwd(qkv, seq, o, dout, lse, H, p, 5, 2, arith=K.GEMM_F16X2).view(B, L, 3, H, 64)
torch.cuda.synchronize()
a, b = res["1"], res["0"]
print("shape", B, L, H, "p", p, "nan in fused:", int(torch.isnan(a).sum()), "nan in two-kernel:", int(torch.isnan(b).sum()))
for part, name in enumerate(("dQ", "dK", "dV")):
    x, y = a[:, :, part], b[:, :, part]
    bad = torch.isnan(x) | ((x - y).abs() > 1e-4 * y.abs().max())
    rows = bad.any(-1).any(-1)            # [B, L]
    if bad.any():
        for bb in range(B):
            r = torch.nonzero(rows[bb]).flatten().tolist()
            if r:
                hh = torch.nonzero(bad[bb].any(0).any(-1)).flatten().tolist()
                dd = torch.nonzero(bad[bb].any(0).any(0)).flatten().tolist()
                print(f"  {name} protein {bb}: {len(r)} bad rows, first {r[:6]} last {r[-3:]}, heads {hh}, d {dd[:4]}..{dd[-2:]} ({len(dd)})")
    else:
        print(f"  {name}: ok, rel {float((x - y).norm() / y.norm()):.2e}")
