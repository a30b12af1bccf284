"""Which path of the AUTO arithmetic carries the gradient error of config 5 (tests/test_gpu_parity_record.py)?
Runs the record's config-5 case once in fp64 and then the device step under several switches."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity_record as T          # noqa: E402
from oracle import batched as obat          # noqa: E402
from oracle import encoder as oenc          # noqa: E402
from protein_transformer_amd import kernels as K_      # noqa: E402
from protein_transformer_amd import synthetic          # noqa: E402
from protein_transformer_amd.models import encoder_only as EO   # noqa: E402
from protein_transformer_amd.protein.Structure import nerf_forward   # noqa: E402
from protein_transformer_amd.train import get_losses   # noqa: E402

cfg, model_s, dm, nl, nh, dff, lens, loss = T.CASES[int(sys.argv[1]) if len(sys.argv) > 1 else 4]
dev = torch.device("cuda:0")
L, B = max(lens), len(lens)
build = lambda ang, seq: nerf_forward(ang.to(dev), seq.to(dev))[0]  # noqa: E731
args = types.SimpleNamespace(loss="drmsd" if loss == "combined" else loss, combined_drmsd_weight=0.5, backbone_loss=False, clip=None)
attempt = int(sys.argv[2]) if len(sys.argv) > 2 else 0
seed = 100 + cfg + 1000 * attempt
batch = synthetic.make_batch(lens, L_pad=L, seed=seed, build_coords=build, frac_missing=0.02)
seq, ang, crd = (batch[k].to(dev) for k in ("seq", "true_ang", "true_crd"))
model = T._make_model(dev, model_s, dm, nl, nh, dff, L, synthetic.angle_means(batch["true_ang"]), seed=7 + cfg + attempt)
params = {k: v.detach().cpu().double() for k, v in model.state_dict().items()}
pe_keys = [k for k in params if k.endswith(".pe")]
leaf = {k: v.clone().requires_grad_() for k, v in params.items() if k not in pe_keys}
pred64 = oenc.encoder_forward({**leaf, **{k: params[k] for k in pe_keys}}, seq.cpu(), nh)
cs = pred64.view(B, L, 12, 2)
rad64 = torch.atan2(cs[..., 1], cs[..., 0])
stats64, crd64, dang64 = obat.batch_loss_and_grads(rad64, seq.cpu(), crd.cpu(), dtype=torch.float64)
rad64.backward(dang64)
ref = {n: v.grad for n, v in leaf.items()}
den = sum(float((ref[n] ** 2).sum()) for n in ref)


def run(label, mode, **flags):
    K_.set_gemm_mode(mode)
    saved = {}
    for k, v in flags.items():
        holder = model if hasattr(model, k) else (EO if hasattr(EO, k) else K_)
        saved[k] = (holder, getattr(holder, k))
        setattr(holder, k, v)
    try:
        model.zero_grad()
        pred = model(seq, ang)
        get_losses(args, pred, ang, crd, seq)
        torch.cuda.synchronize()
        got = {n: p.grad.detach().cpu().double() for n, p in model.named_parameters()}
    finally:
        for k, (holder, v) in saved.items():
            setattr(holder, k, v)
    num = sum(float(((got[n] - ref[n]) ** 2).sum()) for n in ref)
    per = sorted(((float(((got[n] - ref[n]) ** 2).sum()) / float((ref[n] ** 2).sum())) ** 0.5, n) for n in ref if float((ref[n] ** 2).sum()) > 1e-24 * den)
    print(f"{label:44s} total {np.sqrt(num / den):.3e}   worst: " + ", ".join(f"{n.replace('encoder.enc_layers.', 'L')}={e:.1e}" for e, n in per[-3:]),
          flush=True)


run("auto", K_.GEMM_AUTO)
run("auto again", K_.GEMM_AUTO)
run("auto, no side stream", K_.GEMM_AUTO, side_stream_dw=False)
run("auto, no hp forward", K_.GEMM_AUTO, hp_forward=False)
run("auto, no side stream, no hp", K_.GEMM_AUTO, side_stream_dw=False, hp_forward=False)
run("f16x2", K_.GEMM_F16X2)
run("bf16x3", K_.GEMM_BF16X3)
run("bf16x3, no side stream", K_.GEMM_BF16X3, side_stream_dw=False)
run("f32", K_.GEMM_F32)
run("auto, split-K rows off", K_.GEMM_AUTO, SPLIT_K_ROWS=False)

# ---- the products of the top layer's FFN backward in isolation: device result against fp64 on the same operands
cap = []
orig = K_.linear_bwd_input


def spy(dy, w, out=None, **kw):
    r = orig(dy, w, out=out, **kw)
    if len(cap) < 3:
        cap.append((dy.clone(), w.clone(), {k: (v.clone() if torch.is_tensor(v) else v) for k, v in kw.items()}, r.clone()))
    return r


EO.K.linear_bwd_input = spy
K_.set_gemm_mode(K_.GEMM_AUTO)
model.zero_grad()
get_losses(args, model(seq, ang), ang, crd, seq)
torch.cuda.synchronize()
EO.K.linear_bwd_input = orig
for idx, (dy, w, kw, r) in enumerate(cap):
    ref64 = dy.double() @ w.double()
    mag = dy.double().abs() @ w.double().abs()
    gate = kw.get("gate")
    if gate is not None:
        g = 1.0 / (1.0 - kw.get("gate_dropout_p", 0.0))
        ref64 = torch.where(gate > 0, ref64 * g, torch.zeros_like(ref64))
    err = (r.double() - ref64)
    live = dy.abs().amax(1) > 0
    print(f"product {idx}: dy {tuple(dy.shape)} w {tuple(w.shape)} gate={gate is not None} kw={[k for k in kw if k not in ('gate',)]}")
    print(f"   max |err| / sum|x||y| = {(err.abs() / (mag + 1e-300))[live].max().item():.2e}   norm-wise {err.norm().item() / ref64.norm().item():.2e}"
          f"   column sums: {(err.sum(0).norm() / ref64.sum(0).norm()).item():.2e}   rows with dy == 0: {(~live).sum().item()}")
    rowmax = dy.abs().amax(1)
    print(f"   row maxima of dy: min>0 {rowmax[live].min().item():.2e} max {rowmax.max().item():.2e};  per-row error norm / row norm: "
          f"max {(err.norm(dim=1)[live] / ref64.norm(dim=1)[live].clamp_min(1e-300)).max().item():.2e}")
    for mode in (K_.GEMM_BF16X3, K_.GEMM_F16X2):
        kw2 = {k: v for k, v in kw.items() if k not in ("a_scale", "b_scale", "arith")}
        r2 = orig(dy, w, arith=mode, **kw2)
        e2 = r2.double() - ref64
        print(f"   re-run arith={mode} without caller scales: norm-wise {e2.norm().item() / ref64.norm().item():.2e}  column sums {(e2.sum(0).norm() / ref64.sum(0).norm()).item():.2e}")
