"""Find the attention-backward call that produces a non-finite dqkv in the bench workload and describe its inputs."""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench   # noqa: E402
from protein_transformer_amd import kernels as K, synthetic   # noqa: E402
from protein_transformer_amd.optim import FusedSGD   # noqa: E402
from protein_transformer_amd.train import train_step   # noqa: E402

sys.argv = ["bench.py"]
a = bench.parse()
dev = torch.device("cuda:0")
host_batches, angle_means, first = bench.make_batches(a, 0, dev, 2)
resident = [tuple(t.to(dev) for t in b) for b in host_batches]
res_of = [int((b[0] != 20).sum()) for b in host_batches]
torch.manual_seed(synthetic.DEFAULT_SEED)
model = bench.make_model(a, angle_means, dev)
opt = FusedSGD(model, lr=1e-4, weight_decay=10e-3)
args = types.SimpleNamespace(loss=a.loss, combined_drmsd_weight=0.5, backbone_loss=False, clip=1.0)
orig = K.attention_bwd
calls = [0]


def spy(qkv, seq, out, dout, lse, H, p, seed, sid, arith=None, row_scale=None, row_scale_min=None):
    dq = orig(qkv, seq, out, dout, lse, H, p, seed, sid, arith=arith, row_scale=row_scale, row_scale_min=row_scale_min)
    calls[0] += 1
    D = out.shape[1]
    bad = ~torch.isfinite(dq)
    if bad.any() or calls[0] == 1:
        rows = bad.any(dim=1).nonzero().flatten()
        cols = bad.any(dim=0).nonzero().flatten()
        ra = dout.abs().amax(dim=1)
        print(f"call {calls[0]} sid {sid} arith {arith}: non-finite {int(bad.sum())} rows {rows[:8].tolist()}..({len(rows)}) cols {cols[:6].tolist()}..{cols[-3:].tolist()} ({len(cols)})")
        print(f"   dout row amax: min {float(ra.min()):.3e} max {float(ra.max()):.3e} zero rows {int((ra == 0).sum())}; finite dout {bool(torch.isfinite(dout).all())} out {bool(torch.isfinite(out).all())} qkv {bool(torch.isfinite(qkv).all())} lse {bool(torch.isfinite(lse).all())}")
        print(f"   qkv amax {float(qkv.abs().max()):.3e}; zero dout rows (first 40): {(ra == 0).nonzero().flatten()[:40].tolist()}")
        if bad.any():
            B, L = seq.shape
            print("   rows as (protein, pos):", [(int(r) // L, int(r) % L) for r in rows[:12]])
            print("   bad per 64-col block:", bad.view(B * L, -1, 64).any(dim=2).sum(dim=0).tolist())
            b0 = int(rows[0]) // L
            sl = slice(b0 * L, (b0 + 1) * L)
            print("   that protein: dout row amax by 32-row tile:", [f"{float(v):.1e}" for v in ra[sl].view(-1, 32).amax(dim=1)])
            print("   seq pads in that protein:", int((seq[b0] == 20).sum()), " lse finite", bool(torch.isfinite(lse[b0]).all()))
            for mode in (K.GEMM_BF16X3, K.GEMM_F32):
                d2 = orig(qkv, seq, out, dout, lse, H, p, seed, sid, arith=mode)
                print("   arith", mode, "finite:", bool(torch.isfinite(d2).all()))
            sys.exit(0)
    return dq


K.attention_bwd = spy
import protein_transformer_amd.models.encoder_only as EO   # noqa: E402
EO.K.attention_bwd = spy
for k in range(3):
    train_step(model, opt, args, *resident[k % 2], n_res=res_of[k % 2])
    torch.cuda.synchronize()
print("no non-finite dqkv")
