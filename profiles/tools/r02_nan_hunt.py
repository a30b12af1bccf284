"""Run the bench workload step by step and report the first step after which a parameter, a gradient or the loss is not
finite (modes switched in the order bench.py uses).   python profiles/tools/r02_nan_hunt.py [steps]"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench   # noqa: E402
from protein_transformer_amd import kernels, synthetic   # noqa: E402
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
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
modes = [(kernels.GEMM_AUTO, 70), (kernels.GEMM_BF16X3, 24), (kernels.GEMM_F32, 24), (kernels.GEMM_AUTO, 1000)]
k = 0
for mode, n in modes:
    model.gemm_mode = mode
    for _ in range(n):
        if k >= nsteps:
            break
        out = train_step(model, opt, args, *resident[k % 2], n_res=res_of[k % 2])
        flat, g = model.flat_parameters()
        okp, okg = bool(torch.isfinite(flat).all()), bool(torch.isfinite(g).all())
        loss = {kk: float(v) for kk, v in out.items() if torch.is_tensor(v) and v.numel() == 1} if isinstance(out, dict) else str(type(out))
        if not (okp and okg) or k % 25 == 0:
            print(f"step {k} mode {mode}: params finite {okp} grads finite {okg} |g|max {float(g.abs().max()):.3e} loss {loss}", flush=True)
        if not (okp and okg):
            for name, (off, shape) in model._layout.items():
                import numpy as np
                sl = g[off:off + int(np.prod(shape))]
                if not bool(torch.isfinite(sl).all()):
                    print("  non-finite gradient:", name, tuple(shape), "count", int((~torch.isfinite(sl)).sum()))
            sys.exit(0)
        k += 1
print("no non-finite value in", k, "steps")
