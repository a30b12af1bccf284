"""Where the host time of a launch-bound step goes (config 1: d64, 2 layers, 4 proteins): cProfile over 200 steps.
python profiles/tools/r03_host_profile.py"""
import cProfile, os, pstats, sys, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import synthetic
from protein_transformer_amd.models.encoder_only import EncoderOnlyTransformer
from protein_transformer_amd.optim import FusedSGD
from protein_transformer_amd.protein.Sequence import VOCAB
from protein_transformer_amd.protein.Structure import nerf_forward
from protein_transformer_amd.train import train_step
dev = torch.device("cuda:0")
build = lambda ang, seq: nerf_forward(ang.to(dev), seq.to(dev))[0]
b = synthetic.make_batch([64, 40, 33, 20], L_pad=64, seed=1, build_coords=build)
seq, ang, crd = (b[k].to(dev) for k in ("seq", "true_ang", "true_crd"))
model = EncoderOnlyTransformer(2, 8, 64, 128, 64, VOCAB, synthetic.angle_means(b["true_ang"]), True, dropout=0.1).to(dev).train()
opt = FusedSGD(model, lr=1e-4, weight_decay=10e-3)
args = types.SimpleNamespace(loss="drmsd", combined_drmsd_weight=0.5, backbone_loss=False, clip=1.0)
for _ in range(20):
    train_step(model, opt, args, seq, ang, crd, n_res=157)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(200):
    train_step(model, opt, args, seq, ang, crd, n_res=157)
torch.cuda.synchronize()
print(f"un-profiled: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    train_step(model, opt, args, seq, ang, crd, n_res=157)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
