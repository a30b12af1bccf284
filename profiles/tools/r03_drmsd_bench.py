"""dRMSD loss kernels at the benchmark size (32 proteins x 512 residues): forward + backward and forward only.
python profiles/tools/r03_drmsd_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import synthetic
from protein_transformer_amd.losses import drmsd_forward_backward
from protein_transformer_amd.protein.Structure import nerf_forward
dev = torch.device("cuda:0")
build = lambda ang, seq: nerf_forward(ang.to(dev), seq.to(dev))[0]
b = synthetic.make_batch([512] * 32, seed=1, build_coords=build)
seq, crd = b["seq"].to(dev), b["true_crd"].to(dev)
pred = nerf_forward(b["start_ang_rad"].to(dev), seq)[0]
for grad in (True, False):
    for _ in range(3):
        st, g = drmsd_forward_backward(pred, crd, seq, need_grad=grad)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        st, g = drmsd_forward_backward(pred, crd, seq, need_grad=grad)
    e1.record(); torch.cuda.synchronize()
    print(f"lib {os.environ.get('PTAMD_LIB_TAG', 'product')}: drmsd need_grad={grad}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us   checksum {float(st[:, 0].sum()):.6f} "
          f"{float(g.abs().sum()) if g is not None else 0:.6f}")
