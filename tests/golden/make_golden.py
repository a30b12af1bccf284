#!/usr/bin/env python
"""Generate the golden vectors in tests/golden/*.npz by IMPORTING the reference.

Runs only in the build container (needs /root/reference, see refshim.py).
Inputs and expected outputs are stored; nothing of the reference's source is.
Vector sets:
  G1 nerf            G2 generate_coords      G3 drmsd / pairwise distances
  G4 drmsd_work      G5 compute_batch_drmsd  G6 encoder forward
  G7 one train step  G8 mse_over_angles / combine   G9 dataset / batching
  G10 conv-enc       G11 PDB writer          G12 backbone / SOS-EOS selectors

    python tests/golden/make_golden.py
"""
import argparse
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()
warnings.filterwarnings("ignore")

from protein_transformer import dataset as ref_dataset  # noqa: E402
from protein_transformer import losses as ref_losses  # noqa: E402
from protein_transformer import train as ref_train  # noqa: E402
from protein_transformer.protein.Sequence import VOCAB  # noqa: E402
from protein_transformer.protein.Structure import generate_coords, nerf  # noqa: E402

AA = "ACDEFGHIKLMNPQRSTVWY"
N_SC = [1, 2, 4, 5, 7, 0, 6, 4, 5, 4, 4, 4, 3, 5, 7, 2, 3, 3, 10, 8]
CPU = torch.device("cpu")


def ids(s):
    return torch.tensor([VOCAB._char2int[c] for c in s])


def realistic_angles(rng, L):
    """Radians [L,12]: helix/sheet phi-psi mixture, omega ~ pi, bond angles, uniform chi."""
    a = np.zeros((L, 12))
    helix = rng.random(L) < 0.5
    a[:, 0] = np.where(helix, -1.0, -2.1) + rng.normal(0, 0.3, L)
    a[:, 1] = np.where(helix, -0.8, 2.4) + rng.normal(0, 0.3, L)
    om = np.pi + rng.normal(0, 0.05, L)
    a[:, 2] = (om + np.pi) % (2 * np.pi) - np.pi
    a[:, 3] = rng.normal(1.94, 0.03, L)
    a[:, 4] = rng.normal(2.03, 0.03, L)
    a[:, 5] = rng.normal(2.13, 0.03, L)
    a[:, 6:] = rng.uniform(-np.pi, np.pi, (L, 6))
    return a.astype(np.float32)


def nan_unused(crd, seq_ids):
    """[L*14,3] -> NaN in the slots the residue type does not have."""
    crd = crd.copy().reshape(-1, 14, 3)
    for i, r in enumerate(seq_ids):
        crd[i, 4 + N_SC[int(r)]:] = np.nan
    return crd.reshape(-1, 3)


def g1(rng):
    out = {}
    P = 10
    a = rng.normal(0, 3, (P, 3)).astype(np.float32)
    b = a + rng.normal(0, 1.2, (P, 3)).astype(np.float32)
    c = b + rng.normal(0, 1.2, (P, 3)).astype(np.float32)
    l = rng.uniform(1.2, 1.9, P).astype(np.float32)
    th = rng.uniform(-3.1, 3.1, P).astype(np.float32)
    chi = rng.uniform(-6.0, 6.0, P).astype(np.float32)
    l[:4] = [1.379, 1.442, 1.498, 1.229]
    th[3] = 2.0944
    res = [nerf(torch.tensor(a[i]), torch.tensor(b[i]), torch.tensor(c[i]), torch.tensor(l[i]),
                torch.tensor(th[i]), torch.tensor(chi[i])).numpy() for i in range(P)]
    out.update(a=a, b=b, c=c, l=l, theta=th, chi=chi, d=np.stack(res))
    # python-float bond length (backbone call style, StructureBuilder.py:175)
    res2 = [nerf(torch.tensor(a[i]), torch.tensor(b[i]), torch.tensor(c[i]), float(l[i]),
                 torch.tensor(th[i]), torch.tensor(chi[i])).numpy() for i in range(P)]
    out["d_pyfloat_l"] = np.stack(res2)
    # KA1 of SURVEY appendix F
    out["ka1"] = nerf(torch.tensor([0, 0, .001]), torch.tensor([1.442, 0, .001]), torch.tensor([2.0, 1.39, .001]),
                      1.229, torch.tensor(2.0944), torch.tensor(0.5)).numpy()
    return out


def g2(rng):
    out = {}
    cases = [("GA", "uniform")]
    for r in range(20):
        cases.append((AA[r:] + AA[:r], "uniform" if r % 2 == 0 else "realistic"))
    for L, kind in ((8, "uniform"), (64, "realistic"), (64, "uniform"), (128, "realistic")):
        cases.append(("".join(AA[i] for i in rng.integers(0, 20, L)), kind))
    for n, (s, kind) in enumerate(cases):
        L = len(s)
        ang = realistic_angles(rng, L) if kind == "realistic" else rng.uniform(-3, 3, (L, 12)).astype(np.float32)
        crd = generate_coords(torch.tensor(ang), ids(s), CPU).numpy()
        out[f"seq{n}"] = np.array(s)
        out[f"ang{n}"] = ang
        out[f"crd{n}"] = crd
    out["n"] = np.array(len(cases))
    # KA2 of SURVEY appendix F ("GAS")
    ang = np.zeros((3, 12), np.float32)
    for r in range(3):
        for k in range(12):
            ang[r, k] = 0.1 * (k + 1) * (1 if (k + r) % 2 == 0 else -1) + 0.05 * r
    ang[:, 3:6] = (1.94, 2.03, 2.13)
    out["ka2_ang"] = ang
    out["ka2_crd"] = generate_coords(torch.tensor(ang), ids("GAS"), CPU).numpy()
    return out


def g3(rng):
    out = {}
    lits = [
        (np.array([[0, 0, 0], [3, 5, 2], [2, 9, 3]], np.float32), np.array([[0, 0, 0], [9, 3, 1], [4, 7, 8]], np.float32)),
        (np.array([[0., 2.8, 2.95], [2.45, 3.35, 4.4], [4.3, 2., 0.55], [0.9, 3.75, 2.05], [0.35, 3., 1.25]]),
         np.array([[0.75, 4.5, 1.5], [2.85, 4.85, 0.9], [4.65, 1.7, 0.65], [1.55, 1.5, 1.15], [4.4, 1.15, 3.2]])),
        (np.array([[6.1, 0.2, 6.4], [2.2, 4.6, -1.7], [4.5, 2.6, 3.1], [1.1, -0.3, 2.3], [-0.2, 7.3, 3.6]]),
         np.array([[-1.1, 2.5, 6.1], [7.4, -1.6, 6.4], [1.3, 1.2, 1.7], [-0.7, 1.7, 6.], [-0.4, 4.2, 2.9]])),
        ((rng.random((50, 3)) * 10).astype(np.float32), (rng.random((50, 3)) * 10).astype(np.float32)),
        (rng.normal(0, 15, (2000, 3)).astype(np.float32), rng.normal(0, 15, (2000, 3)).astype(np.float32)),
    ]
    for n, (a, b) in enumerate(lits):
        out[f"a{n}"], out[f"b{n}"] = a, b
        out[f"drmsd{n}"] = np.array(ref_losses.drmsd(torch.tensor(a), torch.tensor(b)).item())
    out["n"] = np.array(len(lits))
    p = np.array([[0, 0, 0], [0, 1, 0], [0, 0, 2], [0, 0, 0]], np.float32)
    out["pid_in0"] = p
    out["pid_out0"] = ref_losses.pairwise_internal_dist(torch.tensor(p)).numpy()
    p = np.array([[5.3, -15.2, 300], [-3.3, 234.1, 0]], np.float32)
    out["pid_in1"] = p
    out["pid_out1"] = ref_losses.pairwise_internal_dist(torch.tensor(p)).numpy()
    p = rng.normal(0, 20, (40, 3)).astype(np.float32)
    out["pid_in2"] = p
    out["pid_out2"] = ref_losses.pairwise_internal_dist(torch.tensor(p)).numpy()
    return out


def synth_batch(rng, lens, L_pad, frac_missing=0.0):
    """(seq [B,L] i64 pad 20, pred_ang [B,L,12] rad, true_crd [B,L*14,3] NaN-masked, true_ang_sc [B,L,24])."""
    B = len(lens)
    seq = np.full((B, L_pad), 20, np.int64)
    pred = np.zeros((B, L_pad, 12), np.float32)
    crd = np.zeros((B, L_pad * 14, 3), np.float32)
    tang = np.zeros((B, L_pad, 24), np.float32)
    for b, L in enumerate(lens):
        s = rng.integers(0, 20, L)
        seq[b, :L] = s
        ta = realistic_angles(rng, L)
        pred[b, :L] = ta + rng.normal(0, 0.25, (L, 12)).astype(np.float32)
        pred[b, :L] = (pred[b, :L] + np.pi) % (2 * np.pi) - np.pi
        c = generate_coords(torch.tensor(ta), torch.tensor(s), CPU).numpy()
        c = nan_unused(c, s)
        if frac_missing:
            miss = rng.random(L) < frac_missing
            c.reshape(-1, 14, 3)[miss] = np.nan
        crd[b, :L * 14] = c
        sc = np.stack([np.cos(ta), np.sin(ta)], -1).reshape(L, 24)
        for i, r in enumerate(s):                       # NaN the chi slots the residue does not use
            used = min(N_SC[int(r)], 6)
            sc[i, 12 + 2 * used:] = np.nan
        tang[b, :L] = sc
    return seq, pred, crd, tang


def g4(rng):
    lens = [12, 9, 16, 5]
    seq, pred, crd, _ = synth_batch(rng, lens, 16, frac_missing=0.15)
    out = dict(seq=seq, pred_ang=pred, true_crd=crd)
    for b in range(len(lens)):
        r = ref_losses.drmsd_work(pred[b], crd[b], seq[b], False, True, False)
        out[f"grad{b}"] = r[0].numpy()
        out[f"vals{b}"] = np.array(r[1:], np.float64)
    # tensor call style + KA4 of SURVEY appendix F
    ang = np.zeros((5, 12), np.float32)
    for r_ in range(3):
        for k in range(12):
            ang[r_, k] = 0.1 * (k + 1) * (1 if (k + r_) % 2 == 0 else -1) + 0.05 * r_
    ang[:3, 3:6] = (1.94, 2.03, 2.13)
    s = np.array([5, 0, 15, 20, 20])
    t = generate_coords(torch.tensor(ang[:3] + 0.2), torch.tensor(s[:3]), CPU).numpy()
    t = nan_unused(t, s[:3])
    tc = np.zeros((5 * 14, 3), np.float32)
    tc[:42] = t
    r = ref_losses.drmsd_work(torch.tensor(ang), torch.tensor(tc), torch.tensor(s), False, True, False)
    out.update(ka4_ang=ang, ka4_seq=s, ka4_crd=tc, ka4_grad=r[0].numpy(), ka4_vals=np.array(r[1:], np.float64))
    return out


def tiny_args(loss="drmsd", d_model=32, n_layers=2, n_head=4, dff=64, optimizer="sgd"):
    args = ref_train.create_parser().parse_args([])
    args.model, args.d_model, args.n_layers, args.n_head, args.d_inner_hid = "enc-only", d_model, n_layers, n_head, dff
    args.loss, args.optimizer, args.dropout = loss, optimizer, 0.0
    args.backbone_loss = False
    return args


def make_ref_model(args, angle_means, rng):
    torch.manual_seed(int(rng.integers(0, 2 ** 31)))
    model = ref_train.make_model(args, CPU, angle_means)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    # the reference initialises the output weight to 0; perturb so the encoder matters
    with torch.no_grad():
        model.output_projection.weight.normal_(0, 0.02)
    return model


def g567(rng):
    out = {}
    lens = [14, 10, 16, 7]
    seq, _, crd, tang = synth_batch(rng, lens, 16, frac_missing=0.1)
    angle_means = np.nanmean(np.concatenate([tang[b, :L] for b, L in enumerate(lens)]), axis=0)
    out.update(seq=seq, true_crd=crd, true_ang=tang, angle_means=angle_means, lens=np.array(lens))
    t_seq, t_crd, t_ang = torch.tensor(seq), torch.tensor(crd), torch.tensor(tang)

    args = tiny_args("drmsd")
    model = make_ref_model(args, angle_means, rng)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    for k, v in sd.items():
        if not k.endswith(".pe"):          # the sinusoid table is recomputed by the loader
            out["sd/" + k] = v.numpy()
    out["max_seq_len"] = np.array(sd["encoder.positional_enc.pe"].shape[1])
    out["nhead"] = np.array(args.n_head)

    # G6: forward (eval and train mode with dropout 0 are identical)
    model.eval()
    with torch.no_grad():
        out["g6_pred_eval"] = model(t_seq).numpy()
    model.train()
    out["g6_pred_train"] = model(t_seq).detach().numpy()

    # G5: compute_batch_drmsd with backward through the model
    model.zero_grad()
    pred = model(t_seq)
    vals = ref_losses.compute_batch_drmsd(pred, t_crd, t_seq, do_backward=True)
    out["g5_vals"] = np.array(vals, np.float64)
    for k, p in model.named_parameters():
        out["g5_grad/" + k] = p.grad.numpy().copy()

    # G7: one full step for each loss / optimizer
    for loss, opt_name in (("drmsd", "sgd"), ("combined", "sgd"), ("mse", "sgd"), ("drmsd", "adam"), ("lndrmsd", "sgd")):
        a = tiny_args(loss, optimizer=opt_name)
        a.learning_rate = 1e-2          # big enough that fp32 deltas are well above rounding
        model = ref_train.make_model(a, CPU, angle_means)
        model.load_state_dict(sd)
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        wd = 10e-3
        if opt_name == "adam":
            opt = torch.optim.Adam(model.parameters(), betas=(0.9, 0.98), eps=1e-09, lr=a.learning_rate, weight_decay=wd)
        else:
            opt = torch.optim.SGD(model.parameters(), lr=a.learning_rate, weight_decay=wd)
        model.train()
        opt.zero_grad()
        pred = model(t_seq, t_ang)
        losses = ref_train.get_losses(a, pred, t_ang, t_crd, t_seq, pool=None)
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), a.clip)
        opt.step()
        tag = f"g7_{loss}_{opt_name}"
        out[tag + "/gradnorm"] = np.array(float(gn))
        out[tag + "/lr"] = np.array(a.learning_rate)
        for k in ("loss", "drmsd-full", "lndrmsd-full", "drmsd-bb", "lndrmsd-bb", "combined-full", "mse-full", "mse-bb", "mse-sc"):
            out[tag + "/loss/" + k] = np.array(float(losses[k]))
        # a spread of tensors in full + a norm of the update for every tensor
        keep = ("encoder.input_embedding.emb.weight", "encoder.enc_layers.0.self_attn.wq.weight",
                "encoder.enc_layers.0.self_attn.wo.bias", "encoder.enc_layers.1.pwff.layer1.weight",
                "encoder.enc_layers.1.pwff.layer2.bias", "encoder.enc_layers.0.sublayer_connections.1.norm.weight",
                "output_projection.weight", "output_projection.bias")
        for k, v in model.state_dict().items():
            if k.endswith(".pe"):
                continue
            if k in keep:
                out[tag + "/sd/" + k] = v.numpy().copy()
            out[tag + "/dnorm/" + k] = np.array(float((v - sd[k]).double().norm()))
    return out


def g10(rng):
    """conv-enc with real Conv1d layers (convolutional_encoder.py), embedding and one-hot variants."""
    from protein_transformer.models.convolutional_encoder import ConvEncoderOnlyTransformer
    out = {}
    lens = [14, 9, 16]
    seq, _, _, tang = synth_batch(rng, lens, 16)
    am = np.nanmean(np.concatenate([tang[b, :L] for b, L in enumerate(lens)]), axis=0)
    out.update(seq=seq, angle_means=am)
    w = rng.normal(0, 1, (len(lens), 16, 24)).astype(np.float32)
    out["w"] = w
    for tag, kw in (("emb", dict(conv_kernel_sizes=[3, 5, 3], conv_dim_reductions=[2, 2, 2], use_embedding=True,
                                 conv_out_matches_dm=True)),
                    ("onehot", dict(conv_kernel_sizes=[3, 5], conv_dim_reductions=[0.5, 1], use_embedding=False,
                                    conv_out_matches_dm=True)),
                    # `-m conv-enc-linear-out` (train.py:289-298): no tanh, bias initialised to the angle means themselves
                    ("linear", dict(conv_kernel_sizes=[5], conv_dim_reductions=[1], use_embedding=True,
                                    conv_out_matches_dm=True))):
        torch.manual_seed(int(rng.integers(0, 2 ** 31)))
        model = ConvEncoderOnlyTransformer(nlayers=1, nhead=4, dmodel=32, dff=64, max_seq_len=500, vocab=VOCAB,
                                           angle_means=am, use_tanh_out=tag != "linear", dropout=0.0, **kw)
        if tag == "linear":
            out["linear/init_bias"] = model.output_projection.bias.detach().numpy().copy()
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        with torch.no_grad():
            model.output_projection.weight.normal_(0, 0.05)
        for k, v in model.state_dict().items():
            if not k.endswith(".pe"):
                out[f"{tag}/sd/{k}"] = v.numpy().copy()
        pred = model(torch.tensor(seq))
        out[f"{tag}/pred"] = pred.detach().numpy()
        (pred * torch.tensor(w)).sum().backward()
        for k, p in model.named_parameters():      # the front end + a sample of the rest (G5 covers the layers)
            if any(t in k for t in ("conv_layers", "input_embedding", "output_projection", "wq.weight", "norm.bias")):
                out[f"{tag}/grad/{k}"] = p.grad.numpy().copy()
        out[f"{tag}/kernels"] = np.array(kw["conv_kernel_sizes"])
        out[f"{tag}/reducs"] = np.array(kw["conv_dim_reductions"], dtype=np.float64)
    return out


def g8(rng):
    out = {}
    lens = [9, 6, 12]
    _, _, _, tang = synth_batch(rng, lens, 12)
    pred = np.tanh(rng.normal(0, 1, tang.shape)).astype(np.float32)
    out.update(pred=pred, true=tang)
    tp, tt = torch.tensor(pred), torch.tensor(tang)
    out["full"] = ref_losses.mse_over_angles(tp, tt).numpy()
    out["bb"] = ref_losses.mse_over_angles(tp, tt, bb_only=True).numpy()
    out["sc"] = ref_losses.mse_over_angles(tp, tt, sc_only=True).numpy()
    cases = np.array([(0.01, 0.3, 0.5, 1, 1), (0.01, 0.6, 0, 1, 1), (0.02, 0.3, 1, 1, 1), (0.02, 0.3, 1, 0.02, 1),
                      (0.0571622, 0.0123, 0.5, 0.02, 0.01)])
    out["combine_in"] = cases
    out["combine_out"] = np.array([ref_losses.combine_drmsd_mse(*c, log=False) for c in cases])
    ang = torch.tensor(np.tanh(rng.normal(0, 1, (2, 5, 24))).astype(np.float32))
    out["itt_in"] = ang.numpy()
    out["itt_out"] = ref_losses.inverse_trig_transform(ang).numpy()
    return out


def g9(rng):
    out = {}
    seqs = ["A" * 5, "C" * 10, "D" * 10, "E" * 21, "F" * 40, "G" * 41, "H" * 77, "I" * 120]
    angs = [rng.random((len(s), 24)).astype(np.float32) for s in seqs]
    crds = [rng.random((len(s) * 14, 3)).astype(np.float32) for s in seqs]
    ds = ref_dataset.BinnedProteinDataset(seqs=seqs, angs=angs, crds=crds, add_sos_eos=False,
                                          skip_missing_residues=False, bins="auto")
    out["lens"] = np.array(ds.lens)
    out["hist_bins"] = np.array(ds.hist_bins)
    out["hist_counts"] = np.array(ds.hist_counts)
    out["bin_probs"] = np.array(ds.bin_probs)
    out["bin_map_keys"] = np.array(sorted(ds.bin_map))
    for k, v in ds.bin_map.items():
        out[f"bin_map_{k}"] = np.array(v)
    for n, (s, a, c) in enumerate(zip(seqs, angs, crds)):
        out[f"seq{n}"], out[f"ang{n}"], out[f"crd{n}"] = np.array(s), a, c
    out["n"] = np.array(len(seqs))
    batch = ref_dataset.paired_collate_fn([ds[i] for i in (1, 3, 0)])
    out["collate_seq"], out["collate_ang"], out["collate_crd"] = (t.numpy() for t in batch)
    for opt_cpu in (False, True):
        sampler = ref_dataset.SimilarLengthBatchSampler(ds, 4, dynamic_batch=200, optimize_batch_for_cpus=opt_cpu)
        sampler.cpu_count = 2
        np.random.seed(7)
        sizes = [len(b) for b in sampler]
        out[f"sampler_sizes_cpuopt{int(opt_cpu)}"] = np.array(sizes)
        out[f"sampler_len_cpuopt{int(opt_cpu)}"] = np.array(len(sampler))
    np.random.seed(7)
    sampler = ref_dataset.SimilarLengthBatchSampler(ds, 4, dynamic_batch=200, optimize_batch_for_cpus=False)
    out["sampler_first_batch"] = np.array(next(iter(sampler)))
    pds = ref_dataset.ProteinDataset(seqs=seqs, angs=angs, crds=crds, add_sos_eos=False, skip_missing_residues=False)
    out["pds_order_lens"] = np.array([len(pds[i][0]) for i in range(len(pds))])
    return out


def g11(rng):
    """PDB_Creator.save_pdb on structures built by the reference's own NeRF builder: inputs + the text it writes."""
    import tempfile
    from protein_transformer.protein.PDB_Creator import PDB_Creator
    from protein_transformer.protein.Structure import generate_coords
    from protein_transformer.protein.Sequence import VOCAB as RV
    out = {}
    seqs = ["ACDEFGHIKLMNPQRSTVWY", "GGAWKPY", "MKV"]
    for i, s in enumerate(seqs):
        ids = np.array([RV._char2int[c] for c in s], dtype=np.int64)
        ang = rng.uniform(-np.pi, np.pi, (len(s), 12)).astype(np.float32)
        crd = generate_coords(torch.tensor(ang), torch.tensor(ids), torch.device("cpu")).detach().numpy()
        if i == 1:
            crd = crd.copy()
            crd[14 + 2] = np.nan          # a missing backbone atom: skipped, numbering continues
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "x.pdb")
            PDB_Creator(crd, seq=s).save_pdb(path, title=f"golden {i}")
            text = open(path).read()
        out[f"seq{i}"] = np.array(s)
        out[f"crd{i}"] = crd.astype(np.float32)
        out[f"pdb{i}"] = np.array(text)
    return out


def g12(rng):
    """The two host-side selectors on the drop-in surface: structure_utils.get_backbone_from_full_coords (with and
    without a batch dimension, inverted) and losses.remove_sos_eos_from_input (default vocabulary: SOS = EOS = unknown id)."""
    from protein_transformer.protein.structure_utils import get_backbone_from_full_coords, get_sidechain_from_full_coords
    out = {}
    crd2 = rng.normal(0, 5, (7 * 14, 3)).astype(np.float32)
    crd3 = rng.normal(0, 5, (3, 5 * 14, 3)).astype(np.float32)
    out.update(crd2=crd2, crd3=crd3)
    out["bb2"] = get_backbone_from_full_coords(crd2)
    out["bb3"] = get_backbone_from_full_coords(crd3)
    out["sc2"] = get_sidechain_from_full_coords(crd2)
    out["sc3"] = get_backbone_from_full_coords(crd3, invert=True)
    out["bb2_torch"] = get_backbone_from_full_coords(torch.tensor(crd2)).numpy()
    seqs = [[0, 5, 7, 19], [21, 5, 7, 19], [0, 5, 7, 21], [21, 5, 7, 21], [21, 21, 3, 21]]
    out["sos_ids"] = np.array([VOCAB.sos_id, VOCAB.eos_id])
    for i, s in enumerate(seqs):
        out[f"seq{i}"] = np.array(s)
        out[f"stripped{i}"] = ref_losses.remove_sos_eos_from_input(torch.tensor(s)).numpy()
    out["n"] = np.array(len(seqs))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE)
    ns = ap.parse_args()
    torch.set_num_threads(1)
    for name, fn, seed in (("g1_nerf", g1, 1), ("g2_coords", g2, 2), ("g3_drmsd", g3, 3), ("g4_drmsd_work", g4, 4),
                           ("g567_model_step", g567, 5), ("g8_mse", g8, 8), ("g9_dataset", g9, 9), ("g10_convenc", g10, 10),
                           ("g11_pdb", g11, 11), ("g12_selectors", g12, 12)):
        data = fn(np.random.default_rng(seed))
        path = os.path.join(ns.out, name + ".npz")
        np.savez_compressed(path, **data)
        print(f"{name}: {len(data)} arrays, {os.path.getsize(path) / 1024:.1f} KB")


if __name__ == "__main__":
    main()
