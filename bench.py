#!/usr/bin/env python
"""Headline benchmark: train-step residues/s of the protein-transformer hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--global-batch 32]

Default workload = BASELINE.json configs[3], the one its metric is quoted on: `-m enc-only -dm 512 -nl 6 -nh 8
-dih 2048 -l drmsd`, SGD lr 1e-4 wd 0.01 clip 1, dropout 0.1 ON, 32 synthetic proteins of L=512 per GPU (weak scaling:
the global batch is 32 x N; `--global-batch B` fixes the global batch instead = strong scaling, B / N proteins per GPU).
`--config 1|2|3|5` selects the other BASELINE configurations (parity-test cases; measured for the record, not the
headline).  One step = zero_grad, forward, NeRF + dRMSD loss + backward, gradient all-reduce, clip, optimizer step
(train.train_step = reference train.py:36-46).  Prints ONE JSON line on rank 0 with

  value        - residues/s of the whole job, the MEDIAN of `--passes` (3) timed passes of K steps each, every step's batch
                 uploaded from pinned host memory inside the step by the product's own dataset.DevicePrefetcher (side
                 stream, one batch ahead: SURVEY 8d puts the upload in the step); `resident` = one more pass with the
                 batches already in HBM; `passes` = every pass and the spread;
  roofline     - the DOMINANT KERNEL (largest share of the step's GEMM time; HIP events around every GEMM call on its
                 launch stream in a pass of its own): `achieved` = algorithmic fp32 FLOP of its launches / their measured
                 time, `peak` = the dense peak of the matrix pipe it runs on, `frac` = achieved / peak; beside it
                 `mfma_issue_frac` (x the matrix-pipe products the arithmetic spends per fp32 product) and
                 `emulation_ceiling_frac` (against peak / products), the same three for the whole GEMM family, and the
                 per-kernel table (launches per step, average microseconds) to hold against profiles/;
  strong_scaling - SURVEY 8(e)'s partitioning (global batch 32 -> 32 / N proteins per GPU): at N > 1 the same job timed with
                 that split; at N = 1 the per-GPU steps of N = 2, 4, 8 (16, 8, 4 proteins) and the ceiling they imply;
  arithmetic_modes - ms/step of the same workload, measured in THIS run, with every GEMM / attention in bf16x3 and in
                 the exact-f32 MFMA arithmetic (the strictly fp32-grade alternatives to the default AUTO policy);
  parity       - the metric's second half, "dRMSD-loss delta vs ref", at the size of the line (N = 1): one dropout-0 train_step
                 of the HIP path in its steady-state arithmetic against the oracle step `cpu_baseline` has just timed, same
                 proteins and weights: drmsd_rel, lndrmsd_abs, gradnorm_rel, grad_rel_l2, update_rel_l2;
  cpu_baseline - the CPU oracle (a port of the reference's --no_cuda path) on a bounded sample of the same workload on
                 the host cores of this box, run the way the reference runs: torch.set_num_threads(1) in the main
                 process + a spawn Pool of loss workers (train.py:344,360-365; losses.py:144-147), plus the
                 --sequential_drmsd_loss leg for a per-core figure.
"""
import argparse
import gc
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md: dense f32 matrix peak
BF16_MFMA_PEAK_TFLOPS = 2500.0        # same table: dense bf16 / f16 matrix peak (the marketing figure includes 2:1 sparsity)
TRAFFIC_RECORD = "r06/r06_gemm_hbm_traffic.json"     # under profiles/: PMC traffic of the GEMM launches of the default command

CONFIGS = {   # BASELINE.json configs[i-1]
    # (-dih is not named by BASELINE configs[0]: the reference's default 2048, train.py:479 - 565,272 parameters)
    1: dict(model="enc-only", d_model=64, n_layers=2, n_head=8, d_ff=2048, batch=4, length=64, loss="drmsd", ragged="short"),
    2: dict(model="enc-only", d_model=256, n_layers=4, n_head=8, d_ff=2048, batch=16, length=256, loss="drmsd"),
    3: dict(model="conv-enc|3,7,11|2,2,2", d_model=256, n_layers=6, n_head=8, d_ff=2048, batch=32, length=512, loss="combined"),
    4: dict(model="enc-only", d_model=512, n_layers=6, n_head=8, d_ff=2048, batch=32, length=512, loss="drmsd"),
    5: dict(model="enc-only", d_model=512, n_layers=6, n_head=8, d_ff=2048, batch=8, length=1500, loss="lndrmsd", ragged="binned"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--passes", type=int, default=5, help="timed passes of --steps steps each; the line reports the median one")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling block (SURVEY 8e: global batch / N per GPU)")
    ap.add_argument("--config", type=int, default=4, choices=sorted(CONFIGS), help="BASELINE.json configuration (1-based)")
    ap.add_argument("--batch", type=int, default=None, help="proteins per GPU (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=None, help="proteins of the whole job (strong scaling)")
    ap.add_argument("--length", type=int, default=None)
    ap.add_argument("--d_model", type=int, default=None)
    ap.add_argument("--n_layers", type=int, default=None)
    ap.add_argument("--n_head", type=int, default=None)
    ap.add_argument("--d_ff", type=int, default=None)
    ap.add_argument("--loss", default=None)
    ap.add_argument("--optimizer", default="sgd")
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-proteins", type=int, default=None, help="proteins in the bounded CPU-baseline sample (pool leg)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--verify-dp", action="store_true",
                    help="pre-pass: the SUM over the ranks of the all-reduced gradient against the gradient of the whole global "
                         "batch computed by rank 0 alone (dropout 0); reported as communication.verify_dp")
    ap.add_argument("--no-mode-sweep", action="store_true", help="skip the bf16x3 / f32 re-runs of the timed loop")
    ap.add_argument("--no-attn-row-scales", action="store_true", help="ablation: dqkv row scales by a pass over dqkv")
    ap.add_argument("--no-side-stream", action="store_true", help="ablation: weight-gradient products of small batches on the main stream")
    ap.add_argument("--no-hp-forward", action="store_true", help="ablation: the FFN-layer-1 forward product on ptamd_gemm instead of ptamd_gemm_hp")
    ap.add_argument("--no-hp-qkv", action="store_true", help="ablation: the QKV product on ptamd_gemm instead of ptamd_gemm_hp")
    ap.add_argument("--no-attn-keep-bits", action="store_true", help="ablation: the fused attention backward kernel draws the dropout decisions again instead of reading the forward kernel's")
    ap.add_argument("--no-ffn-gate-mask", action="store_true", help="ablation: the gated dX product of FFN layer 2 reads the fp32 activation instead of its 1-bit gate")
    ap.add_argument("--no-top-layer-scales", action="store_true", help="ablation: the top layer's FFN weight-gradient products in bf16x3 (no pass over its dy2)")
    ap.add_argument("--dw-group", default="auto", choices=["auto", "pairs", "layer", "off"],
                    help="grouping of the weight-gradient products of a layer (ptamd_gemm_group); off = one by one (ablation)")
    ap.add_argument("--no-kv-planes", action="store_true", help="ablation: the QKV product stores K / V as fp32 and the attention kernels scale, split and stage them themselves (rounds 2-4)")
    ap.add_argument("--no-weights-prep", action="store_true", help="ablation: scales / bounds / planes of the weights by the separate launches of rounds 2-4 in front of every forward pass instead of inside the optimizer step (csrc/wprep.hip)")
    ap.add_argument("--no-hp-dx", action="store_true", help="ablation: dX of FFN layer 2 on the staging kernel instead of ptamd_gemm_hp")
    ap.add_argument("--attn-mode", default=None, choices=["f32", "bf16x3", "f16x2"],
                    help="arithmetic of the attention kernels alone (ablation; default: that of --gemm-mode)")
    ap.add_argument("--gemm-mode", default="auto", choices=["f32", "bf16x3", "bf16x3full", "f16x2", "auto"],
                    help="arithmetic of the encoder GEMMs and attention (include/ptamd.h: PTAMD_GEMM_*)")
    a = ap.parse_args()
    cfg = dict(CONFIGS[a.config])
    for k in ("batch", "length", "d_model", "n_layers", "n_head", "d_ff", "loss"):
        if getattr(a, k) is None:
            setattr(a, k, cfg[k])
    a.model, a.ragged = cfg["model"], cfg.get("ragged")
    return a


# ----------------------------------------------------------------------------- CPU baseline (the checker, timed)
def _cpu_leg(a, batch_cpu, params, n, pool, keep=False):
    """One timed oracle step on n proteins of the batch.  The oracle keeps the reference's assertion on the bond angle
    (Structure.py:42, theta in [-pi, pi] against the DOUBLE pi): a float32 angle within 9e-8 below pi rounds to
    float32(pi) > pi and trips it - about once in 1e7 angles, on the reference as on the oracle.  Such a sample says
    nothing about speed: the leg moves on to the next n proteins of the batch (at most 4 attempts)."""
    from oracle import step as ostep
    avail = batch_cpu["seq"].shape[0]
    last = None
    for attempt in range(4):
        idx = [(attempt * n + i) % avail for i in range(n)]
        trainer = ostep.CpuTrainer({k: v.clone() for k, v in params.items()}, a.n_head, loss=a.loss, optimizer=a.optimizer,
                                   lr=1e-4, clip=1.0, pool=pool)
        seq, ang, crd = (batch_cpu[k][idx] for k in ("seq", "true_ang", "true_crd"))
        try:
            rate, dt = ostep.time_cpu_steps(trainer, (seq, ang, crd), n_steps=1, keep_grads=keep)
            return rate, dt, (trainer, idx)
        except AssertionError as e:                      # the reference's own input assertion, see above
            last = e
    raise last


def cpu_baseline(a, batch_cpu, params):
    """One step of the CPU oracle the way the reference's --no_cuda path runs it: single-threaded torch in the main
    process (train.py:344) and the per-protein loss fanned out over a spawn Pool (train.py:360-365, losses.py:144-147);
    then one step with the sequential loss (--sequential_drmsd_loss) for a per-core figure.  Bounded samples of the
    same workload: the per-protein loss alone takes ~12 s of one core at L = 512."""
    import multiprocessing as mp
    torch.set_num_threads(1)
    host_cores = os.cpu_count() or 1
    avail, L = batch_cpu["seq"].shape
    heavy = L * L * a.d_model >= 200 * 200 * 512
    # The pool leg takes the WHOLE batch when the host has a core per protein (the reference's pool maps all of them at
    # once: one protein per worker, ~12 s at L = 512); on a host with fewer cores a bounded sample of max(8, cores) proteins
    # keeps the leg at 10-30 s
    n_pool = a.cpu_proteins or (min(avail, max(8, min(32, host_cores))) if heavy else min(avail, 32))
    workers = max(1, min(host_cores, n_pool))           # the reference asks for cpu_count() workers; only n_pool get work
    t0 = time.perf_counter()
    # (workers pinned to one torch thread each: left at the default every worker starts cpu_count() intra-op threads and
    # the pool leg runs SLOWER than the sequential one on a many-core host - the reference has that problem as it stands)
    with mp.get_context("spawn").Pool(workers, initializer=torch.set_num_threads, initargs=(1,)) as pool:
        pool.map(abs, range(workers))                    # workers up (imports done) before the clock starts
        t_spawn = time.perf_counter() - t0
        # (keep=True: the step's losses and gradients stay behind for the `parity` block - a 76 MB clone inside a ~30 s step)
        rate_pool, dt_pool, oracle_step = _cpu_leg(a, batch_cpu, params, n_pool, pool, keep=True)
    n_seq = 1 if heavy else min(avail, 4)
    rate_seq, dt_seq, _ = _cpu_leg(a, batch_cpu, params, n_seq, None)
    return oracle_step, {"value": round(rate_pool, 2), "unit": "residues/s", "cores": workers, "kind": "port",
            "host_cores": host_cores,
            "sample": f"1 step of oracle.step.CpuTrainer on {n_pool} of the {avail} proteins of a batch (L={L}, same "
                      f"model, dropout 0, torch threads = 1 in the main process and in every worker, loss in a spawn Pool of {workers} "
                      f"workers [the reference asks for cpu_count() = {host_cores}; {n_pool} proteins keep {workers} busy]): "
                      f"{dt_pool:.1f} s (+ {t_spawn:.1f} s pool start-up)",
            "sequential": {"value": round(rate_seq, 2), "unit": "residues/s", "cores": 1,
                           "sample": f"1 step on {n_seq} protein(s), --sequential_drmsd_loss: {dt_seq:.1f} s"}}


def parity_block(a, model, opt, args, batch_cpu, params, oracle_step, dev):
    """The second half of BASELINE.json's metric, "dRMSD-loss delta vs ref", at the size of the line it is printed in: ONE
    dropout-0 `train_step` of the HIP path - the model as the timed steps left it, AUTO arithmetic in its steady state (guard
    trusted), same batch, same weights - against the step the CPU oracle has just been timed on (`cpu_baseline`'s pool leg:
    the reference's arithmetic, fp32 on the host).  Reference quantities: compute_batch_drmsd's returned means over the
    proteins (losses.py:153-172), the gradient clip_grad_norm_ sees and the parameter update of train.py:41-46."""
    from protein_transformer_amd.train import train_step
    trainer, idx = oracle_step
    seq, ang, crd = (batch_cpu[k][idx].to(dev) for k in ("seq", "true_ang", "true_crd"))
    p, pa = model.dropout, model.attn_dropout
    fuse = getattr(opt, "zero_grad_in_step", False)
    model.set_dropout(0.0)
    opt.zero_grad_in_step = False                       # (the gradient must survive the step to be compared)
    try:
        flat, grad = model.flat_parameters()
        w0 = flat.detach().clone()
        losses = train_step(model, opt, args, seq, ang, crd, n_res=int((seq != 20).sum()))
        torch.cuda.synchronize()
        g, w1 = grad.detach().double().cpu(), flat.detach().clone()
    finally:
        model.set_dropout(p, pa)
        opt.zero_grad_in_step = fuse
    # the oracle's tensors in the order of the flat buffer
    def flatten(named, like):
        out = torch.zeros(model._flat_numel, dtype=torch.float64)
        for name, (off, shape) in model._layout.items():
            t = named.get(name)
            if t is not None:
                out[off:off + t.numel()] = t.detach().double().reshape(-1)
        return out
    g_ref = flatten(trainer.last_grads, None)
    dw_ref = flatten({k: v.detach() - params[k] for k, v in trainer.params.items()}, None)
    dw = (w1.double() - w0.double()).cpu()
    ref = {k: float(trainer.last_losses[k]) for k in ("drmsd-full", "lndrmsd-full", "drmsd-bb", "mse-full")}
    hip = {k: float(losses[k]) for k in ref}
    rel = lambda x, y: abs(x - y) / max(abs(y), 1e-300)                                     # noqa: E731
    gn, gn_ref = float(g.norm()), float(trainer.last_grad_norm)
    guard = model.auto_guard
    out = {"drmsd_rel": rel(hip["drmsd-full"], ref["drmsd-full"]),
           "lndrmsd_abs": abs(hip["lndrmsd-full"] - ref["lndrmsd-full"]),
           "drmsd_bb_rel": rel(hip["drmsd-bb"], ref["drmsd-bb"]), "mse_rel": rel(hip["mse-full"], ref["mse-full"]),
           "gradnorm_rel": rel(gn, gn_ref), "grad_rel_l2": float((g - g_ref).norm()) / max(gn_ref, 1e-300),
           "update_rel_l2": float((dw - dw_ref).norm()) / max(float(dw_ref.norm()), 1e-300),
           "hip": {**hip, "grad_norm": gn}, "oracle": {**ref, "grad_norm": gn_ref},
           "n_proteins": int(seq.shape[0]), "seq_len": int(seq.shape[1]),
           "arithmetic": ("auto steady state" if not (guard.off.any() or guard.wide.any()) and guard.measured_steps > 0 else
                          f"auto, {int(guard.off.sum())} sites off their bounds / {int(guard.wide.sum())} products in bf16x3")
                         if (model.gemm_mode in (None, 4)) else f"gemm mode {model.gemm_mode}",
           "tolerance": {"drmsd_rel": 1e-4, "lndrmsd_abs": 1e-6, "grad_rel_l2": 1e-3},
           "what": "one dropout-0 train_step of the HIP path (the model as the timed steps left it) against the CPU oracle's step "
                   "on the same proteins and weights (cpu_baseline's pool leg; the reference's fp32 arithmetic): batch means of "
                   "drmsd / lndrmsd / backbone drmsd (losses.py:153-172), the MSE over angles, the gradient before the clip "
                   "(norm and rel-L2 of the whole vector) and the parameter update (train.py:41-46)"}
    out["ok"] = bool(out["drmsd_rel"] < 1e-4 and out["lndrmsd_abs"] < 1e-6 and out["grad_rel_l2"] < 1e-3)
    return {k: (float(f"{v:.4g}") if isinstance(v, float) else v) for k, v in out.items()}


# ----------------------------------------------------------------------------- workloads
def make_batches(a, rank, dev, n_batches):
    """Synthetic batches on the host (pinned); returns (list of (seq, ang, crd) CPU tensors, angle_means, first batch dict)."""
    from protein_transformer_amd import synthetic
    from protein_transformer_amd.dataset import pack_batch
    from protein_transformer_amd.protein.Structure import nerf_forward
    build = lambda ang, seq: nerf_forward(ang.to(dev), seq.to(dev))[0]            # noqa: E731
    seed0 = synthetic.DEFAULT_SEED + 97 * rank
    if a.ragged == "binned":
        # BASELINE configs[4]: variable lengths (log-normal, median 200, clipped to [20, L]) drawn through the reference's
        # binned-random batching: BinnedProteinDataset -> SimilarLengthBatchSampler (residue budget batch x L per batch)
        from protein_transformer_amd.dataset import BinnedProteinDataset, SimilarLengthBatchSampler, make_paired_collate_fn
        from protein_transformer_amd.protein.Sequence import VOCAB
        rng = np.random.default_rng(seed0)
        lens = sorted(int(x) for x in np.clip(rng.lognormal(np.log(200), 0.8, 256), 20, a.length))
        seqs, angs, crds, first = [], [], [], None
        for i in range(0, len(lens), 64):
            chunk = lens[i:i + 64]
            b = synthetic.make_batch(chunk, L_pad=max(chunk), seed=seed0 + i, build_coords=build)
            first = first or b
            for j, n in enumerate(chunk):
                seqs.append(VOCAB.ints2str(b["seq"][j, :n].tolist()))
                angs.append(b["true_ang"][j, :n].double().numpy())
                crds.append(b["true_crd"][j, :n * 14].double().numpy())
        ds = BinnedProteinDataset(seqs=seqs, angs=angs, crds=crds, add_sos_eos=False, skip_missing_residues=False,
                                  max_seq_len=a.length)
        smp = SimilarLengthBatchSampler(ds, a.batch, dynamic_batch=a.batch * a.length, optimize_batch_for_cpus=False)
        collate = make_paired_collate_fn(a.length)
        np.random.seed(seed0)
        batches = []
        while len(batches) < n_batches:                              # one pass of the sampler = one epoch
            for idx in smp:
                batches.append(collate([ds[int(i)] for i in idx]))
                if len(batches) == n_batches:
                    break
        am = synthetic.angle_means(first["true_ang"])
        return [pack_batch(b, pin=True) for b in batches], am, first
    out, first = [], None
    for i in range(min(n_batches, 2)):
        if a.ragged == "short":                                     # configs[0]: mixed lengths in [16, L]
            lens = list(np.random.default_rng(seed0 + i).integers(16, a.length + 1, a.batch))
            lens[0] = a.length
        else:
            lens = [a.length] * a.batch
        b = synthetic.make_batch([int(x) for x in lens], L_pad=a.length, seed=seed0 + i, build_coords=build)
        first = first or b
        out.append(pack_batch(tuple(b[k] for k in ("seq", "true_ang", "true_crd")), pin=True))       # as the collate function packs them
    return out, synthetic.angle_means(first["true_ang"]), first


def verify_dp(a, model, args, dev):
    """`--verify-dp`: does the data-parallel step compute what one process computes on the global batch?  Every rank runs
    forward + loss + backward on ITS batch with dropout 0 and the gradients are SUM-all-reduced the way a step does it
    (per-layer hooks from the backward pass); rank 0 then regenerates every rank's batch, runs the concatenated global batch
    alone (no collective: dp.single_process) and reports || sum_ranks g - g_full || / || g_full ||.  Reference semantics:
    the gradient is the SUM over proteins (losses.py:166-167)."""
    from protein_transformer_amd import dp
    from protein_transformer_amd.train import get_losses
    world, rank = dp.world_size(), dp.rank()
    p, pa = model.dropout, model.attn_dropout
    model.set_dropout(0.0)

    def grad_of(batch):
        seq, ang, crd = (t.to(dev) for t in batch)
        model.zero_grad()
        pred = model(seq, ang)
        get_losses(args, pred, ang, crd, seq, n_res=int((seq != 20).sum()))
        dp.all_reduce_gradients(model)
        torch.cuda.synchronize()
        return model.flat_parameters()[1].clone()

    g_sum = grad_of(make_batches(a, rank, dev, 1)[0][0])
    out = None
    if rank == 0:
        parts = [make_batches(a, r, dev, 1)[0][0] for r in range(world)]
        Lmax = max(b[0].shape[1] for b in parts)

        def pad(t, n, value):                                          # [B, n_i, ...] -> [B, n, ...]
            if t.shape[1] == n:
                return t
            fill = torch.full((t.shape[0], n - t.shape[1]) + tuple(t.shape[2:]), value, dtype=t.dtype)
            return torch.cat([t, fill], 1)
        full = (torch.cat([pad(b[0], Lmax, 20) for b in parts]), torch.cat([pad(b[1], Lmax, 0.0) for b in parts]),
                torch.cat([pad(b[2], Lmax * 14, 0.0) for b in parts]))
        with dp.single_process(model):
            g_full = grad_of(full)
        num, den = float((g_sum.double() - g_full.double()).norm()), float(g_full.double().norm())
        out = {"rel_l2": num / max(den, 1e-300), "grad_norm_full_batch": den, "global_batch": int(full[0].shape[0]),
               "tolerance": 1e-4, "ok": bool(num <= 1e-4 * den),
               "what": "|| SUM over ranks of the all-reduced gradient - gradient of the whole global batch computed by rank 0 "
                       "alone || / || the latter ||, dropout 0, same model"}
    dp.barrier()
    model.set_dropout(p, pa)
    model.zero_grad()
    return out


def make_model(a, angle_means, dev):
    from protein_transformer_amd.models.convolutional_encoder import ConvEncoderOnlyTransformer
    from protein_transformer_amd.models.encoder_only import EncoderOnlyTransformer
    from protein_transformer_amd.protein.Sequence import VOCAB
    if a.model.startswith("conv-enc"):
        _, ks, rs = a.model.split("|")
        return ConvEncoderOnlyTransformer(a.n_layers, a.n_head, a.d_model, a.d_ff, a.length, VOCAB, angle_means, True,
                                          [int(k) for k in ks.split(",")], [float(r) for r in rs.split(",")], True, True,
                                          dropout=a.dropout).to(dev).train()
    return EncoderOnlyTransformer(a.n_layers, a.n_head, a.d_model, a.d_ff, a.length, VOCAB, angle_means, True,
                                  dropout=a.dropout).to(dev).train()


def main():
    a = parse()
    from protein_transformer_amd import dp, kernels, synthetic
    from protein_transformer_amd.optim import FusedAdam, FusedSGD
    from protein_transformer_amd.dataset import DevicePrefetcher, pack_batch
    from protein_transformer_amd.train import train_step

    dp.init_from_env()
    world, rank = dp.world_size(), dp.rank()
    if world != a.gpus:
        sys.exit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    dev = torch.device("cuda", dp.local_rank())
    torch.cuda.set_device(dev)
    modes = {"f32": kernels.GEMM_F32, "bf16x3": kernels.GEMM_BF16X3, "bf16x3full": kernels.GEMM_BF16X3_FULL,
             "f16x2": kernels.GEMM_F16X2, "auto": kernels.GEMM_AUTO}
    scaling = "weak"
    if a.global_batch is not None:                                   # strong scaling: fixed global batch, B / N per GPU
        if a.global_batch % world:
            sys.exit(f"--global-batch {a.global_batch} is not a multiple of {world} GPUs")
        a.batch, scaling = a.global_batch // world, "strong"

    n_host_batches = a.steps + a.warmup if a.ragged == "binned" else 2
    host_batches, angle_means, first = make_batches(a, rank, dev, n_host_batches)
    resident = [tuple(t.to(dev) for t in b) for b in host_batches]
    res_of = [int((b[0] != 20).sum()) for b in host_batches]

    torch.manual_seed(synthetic.DEFAULT_SEED)
    model = make_model(a, angle_means, dev)
    model.gemm_mode = modes[a.gemm_mode]
    model.attn_mode = None if a.attn_mode is None else modes[a.attn_mode]
    model.attn_row_scales = not a.no_attn_row_scales
    model.side_stream_dw = not a.no_side_stream
    if os.environ.get("PTAMD_DW_SLOTS"):
        kernels.DW_SLOTS = int(os.environ["PTAMD_DW_SLOTS"])
    model.hp_forward = not a.no_hp_forward
    model.hp_qkv, model.hp_dx = not a.no_hp_qkv, not a.no_hp_dx
    model.keep_attn_bits = not a.no_attn_keep_bits
    model.ffn_gate_mask = not a.no_ffn_gate_mask
    model.dw_group = a.dw_group
    model.top_layer_scales = not a.no_top_layer_scales
    model.weights_prep = not a.no_weights_prep
    model.kv_planes = not a.no_kv_planes
    model.dropout_seed += 7919 * rank
    dp.attach(model)
    opt = (FusedAdam(model, lr=1e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=10e-3) if a.optimizer == "adam"
           else FusedSGD(model, lr=1e-4, weight_decay=10e-3))
    opt.zero_grad_in_step = True            # as train.setup_model_optimizer_scheduler sets it for the product's loop (optim.py)
    args = types.SimpleNamespace(loss=a.loss, combined_drmsd_weight=0.5, backbone_loss=False, clip=1.0)
    nb = len(resident)

    def step(i):
        return train_step(model, opt, args, *resident[i % nb], n_res=res_of[i % nb])

    def max_over_ranks(dt):
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())

    def timed(fn, first_step):
        dp.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            out = fn(first_step + i)
        dp.barrier()
        torch.cuda.synchronize()
        return max_over_ranks(time.perf_counter() - t0), out

    shares = {}                                                      # per-GPU shares of the host batches, packed like whole ones

    def timed_upload(first_step, batches=None, proteins=None, steps=None):
        """K steps with every batch coming from pinned host memory INSIDE the step, the way train_epoch gets them
        (train.py: dataset.DevicePrefetcher - the next batch's copy on a side stream under this step, the residue count on
        the host).  `proteins`: only the first so many of every batch (the per-GPU share of a strongly scaled job);
        `steps`: steps of this pass (default: K)."""
        steps = a.steps if steps is None else steps
        src = host_batches if batches is None else batches
        if proteins is not None:
            key = (id(src), proteins)
            if key not in shares:
                shares[key] = [pack_batch((s[:proteins], g[:proteins], c[:proteins]), pin=True) for s, g, c in src]
            src = shares[key]
        feed = (src[(first_step + i) % len(src)] for i in range(steps))
        out, n_res = None, 0
        dp.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for seq, ang, crd, n in DevicePrefetcher(feed, dev):
            out = train_step(model, opt, args, seq, ang, crd, n_res=n)
            n_res += n
        dp.barrier()
        torch.cuda.synchronize()
        return max_over_ranks(time.perf_counter() - t0), out, n_res

    who = dp.describe()                                              # backend, RCCL version, every rank's device identity
    verified = verify_dp(a, model, args, dev) if a.verify_dp else None
    for i in range(a.warmup):
        losses = step(i)
    timing = None if a.no_kernel_timing else []
    if timing is not None:      # GEMM launches per step are counted in the warm-up; events are created up front
        kernels.GEMM_TIMING, kernels.GEMM_EVENT_POOL = [], [torch.cuda.Event(enable_timing=True) for _ in range(1200)]
        step(0)                                                      # the launch count per step does not depend on the batch
        per_step = len(kernels.GEMM_TIMING)
        kernels.GEMM_TIMING = None
        kernels.GEMM_BYTES.clear()
        kernels.GEMM_EVENT_POOL = [torch.cuda.Event(enable_timing=True) for _ in range(2 * per_step * a.steps + 8)]
        for e in kernels.GEMM_EVENT_POOL:
            e.record()          # materialise the underlying hipEvents outside the timed region
    gc.collect()
    gc.disable()                # a generation-2 collection over the event pool costs ~60 ms when it lands in the timed steps
    # THE timed region, `--passes` times: K steps each, nothing else on the host; the line reports the median pass
    passes = []
    for _ in range(max(1, a.passes)):
        d, losses, n_res_timed = timed_upload(a.warmup)
        passes.append(d)
    dt = sorted(passes)[len(passes) // 2]
    dt_res, _ = timed(step, a.warmup)                                # the batches already resident in HBM (rounds 1-4's `value`)
    comm = {k: who[k] for k in ("backend", "rccl_version", "world_size", "ranks_ok", "distinct_devices", "ranks_seen") if k in who}
    if world > 1:
        # one step with the hooks traced: bytes of every slice handed to the all-reduce (in the order of the backward pass:
        # output layer, encoder layers top down, front end) and the stream each reduction was issued from
        dp.hook_trace_start()
        step(0)
        torch.cuda.synchronize()
        trace = dp.hook_trace_stop()
        comm["allreduce_bytes_per_layer"] = [t["bytes"] for t in trace]
        comm["hooks"] = trace
    if world > 1:
        # the same K steps with two events per step around the tail wait of the gradient all-reduce: how much of the
        # reduction the overlap with backward did NOT hide (a pass of its own, like the GEMM events below)
        dp.comm_meter_start()
        dt_comm, _ = timed(step, a.warmup)
        wait_ms, nbytes = dp.comm_meter_stop()
        wt = torch.tensor([wait_ms], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(wt, op=torch.distributed.ReduceOp.MAX)     # the slowest rank's wait
        wait_ms = float(wt.item())
        comm.update({"allreduce_wait_ms": round(wait_ms, 4), "comm_bytes": nbytes,
                "ms_per_step_in_this_pass": round(1e3 * dt_comm / a.steps, 3),
                "what": "per step and rank: time the compute stream waited for the SUM all-reduce of the flat gradient (issued per "
                        "encoder layer from the backward pass, RCCL stream) after backward had been enqueued; bytes all-reduced "
                        "(+ one 19-entry fp64 vector of loss statistics)",
                "reserved_cus": int(kernels.GEMM_RESERVED_CUS)})
    dt_inst = None
    if timing is not None:
        # the same K steps once more with two HIP events around every ptamd_gemm call (150 event records per step cost
        # ~0.4 ms of host time per step, which is why this pass is not the one `value` is taken from)
        # (the weight-gradient products go back to the main stream for this pass: two kernels that share the chip from two
        # streams each report a duration that includes waiting for CUs, and the sum would no longer be GEMM time)
        kernels.GEMM_TIMING = timing
        side, model.side_stream_dw = model.side_stream_dw, False
        dt_inst, _ = timed(step, a.warmup)
        model.side_stream_dw = side
        kernels.GEMM_TIMING = None
    sweep = {}
    if not a.no_mode_sweep and a.gemm_mode == "auto":
        for name in ("bf16x3", "f32"):
            model.gemm_mode = modes[name]
            step(0)
            d, _ = timed(step, a.warmup)
            sweep[name] = {"ms_per_step": round(1e3 * d / a.steps, 3), "residues_per_s": round(world * n_res_timed / d, 1)}
        model.gemm_mode = modes[a.gemm_mode]
        # the default once more, LAST: on launch-bound workloads (configs 1, 2) every later pass of a process runs a few
        # per cent faster than the first ones (host side warm-up) - an order effect that would otherwise read as "AUTO is
        # slower than the arithmetics timed after it" although they run the very same kernels there
        step(0)
        d, _ = timed(step, a.warmup)
        sweep["auto, timed again after the sweep"] = {"ms_per_step": round(1e3 * d / a.steps, 3),
                                                       "residues_per_s": round(world * n_res_timed / d, 1)}
    # the reference's only parallelism is one protein per worker of a GLOBAL batch (losses.py:144-147), i.e. the
    # global batch of the configuration split over the GPUs.  `value` above is weak scaling (the configuration's batch on every
    # GPU); this block times the split: at N > 1 the job itself with batch / N proteins per GPU, at N = 1 the per-GPU step of
    # N = 2, 4, 8 - what a GPU of such a job computes, without its all-reduce: the ceiling of the strong-scaling curve.
    strong = None
    if not a.no_strong and not a.ragged and scaling == "weak":
        full_ms = 1e3 * dt / a.steps
        if world > 1 and a.batch % world == 0:
            pb = a.batch // world
            for i in range(a.warmup):
                timed_upload(i, proteins=pb)
            d, _, n = timed_upload(a.warmup, proteins=pb)
            strong = {"scaling": "strong", "global_batch": a.batch, "proteins_per_gpu": pb, "n_gpus": world,
                      "ms_per_step": round(1e3 * d / a.steps, 3), "value": round(world * n / d, 1), "unit": "residues/s",
                      "what": f"the same job with the configuration's global batch of {a.batch} proteins split over the {world} GPUs "
                              f"(SURVEY 8e), gradient all-reduce included; divide by the N = 1 line's value for the speed-up"}
        elif world == 1:
            per = {}
            for n_gpu in (2, 4, 8):
                if a.batch % n_gpu:
                    continue
                pb = a.batch // n_gpu
                for i in range(a.warmup):
                    timed_upload(i, proteins=pb)
                # the median of five passes of K x N steps each (as many proteins per pass as a pass of `value`): a share of 4
                # proteins is ~135 launches in < 3 ms, and 20-step passes (53 ms) on a box with a busy host read 2.82, 3.07 and
                # 3.82 where other boxes read 2.66 in every pass (profiles/r06/r06_final_box_spread.txt)
                ks = a.steps * n_gpu
                runs = sorted(timed_upload(a.warmup, proteins=pb, steps=ks)[::2] for _ in range(5))
                d, n = runs[2]
                per[str(n_gpu)] = {"proteins_per_gpu": pb, "ms_per_step": round(1e3 * d / ks, 3),
                                   "passes_ms_per_step": [round(1e3 * r[0] / ks, 3) for r in runs], "steps_per_pass": ks,
                                   "residues_per_s_per_gpu": round(n / d, 1),
                                   "speedup_ceiling": round(full_ms / (1e3 * d / ks), 3)}
            strong = {"scaling": "strong", "global_batch": a.batch, "per_gpu_step_at_n_gpus": per,
                      "strong_scaling_ceiling_8gpu": per.get("8", {}).get("speedup_ceiling"),
                      "what": f"ONE GPU running the per-GPU share of the configuration's global batch of {a.batch} proteins at N = 2, 4, "
                              f"8 (SURVEY 8e); speedup_ceiling = ms({a.batch} proteins) / ms(share): what N GPUs could reach "
                              f"if the all-reduce of the 75.75 MB gradient were free"}
            for i in range(a.warmup):                                  # back to the full batch for whatever follows
                step(i)
    gc.enable()

    # a step whose numbers are NaN runs FASTER (the matrix pipe draws less power on constant data): a throughput measured on
    # a diverged model is not a measurement.  Checked on every rank after the last timed loop, fatal.
    flat_p, flat_g = model.flat_parameters()
    if not (bool(torch.isfinite(flat_p).all()) and bool(torch.isfinite(flat_g).all())):
        sys.exit("bench.py: non-finite parameters / gradients after the timed steps - the measurement is invalid")
    # ... and after all those steps every rank must hold the SAME parameters, bit for bit (same SUM-reduced gradient, same
    # update): max - min over the ranks of sum |p| and of a hash of the flat buffer's bit patterns, both 0 or the line says so
    comm["param_checksum_spread"] = dp.param_checksum_spread(model)
    if verified is not None or a.verify_dp:
        comm["verify_dp"] = verified

    # HBM bytes per GEMM launch: NOT measured in this run (PMC counters need rocprofv3 around the process) but read from the
    # record of separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this command (profiles/tools/r04_collect.sh ->
    # profiles/summarize.py traffic); the record names the GEMM sources it was measured on and is refused when they differ
    traffic, traffic_note = None, None
    tpath = os.path.join(ROOT, "profiles", TRAFFIC_RECORD)
    if a.config == 4 and a.batch == 32 and a.gemm_mode == "auto":
        from protein_transformer_amd.build import gemm_source_digest
        if not os.path.exists(tpath):
            traffic_note = f"no record profiles/{TRAFFIC_RECORD}"
        else:
            with open(tpath) as f:
                rec = json.load(f)
            if rec.get("gemm_source_digest") != gemm_source_digest():
                traffic_note = (f"profiles/{TRAFFIC_RECORD} was measured on other GEMM sources (digest "
                                f"{str(rec.get('gemm_source_digest'))[:12]} != {gemm_source_digest()[:12]}): stale, not reported")
            else:
                traffic = round(rec["hbm_bytes_per_launch"])
                traffic_note = (f"read from profiles/{TRAFFIC_RECORD} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                f"command on the same GEMM sources, digest {gemm_source_digest()[:12]}; FETCH x 2 per the gfx950 "
                                f"correction of the guide); not measured in this run")
    roofline = None
    if timing:
        def pipe_peak_of(products):
            return F32_MFMA_PEAK_TFLOPS if products == 1 else BF16_MFMA_PEAK_TFLOPS

        def rates(entries):
            """algorithmic fp32 TF/s, matrix-pipe TF/s issued, the pipe's dense peak and the three fractions of a set of
            timed launches (flop, event, event, products per fp32 product, kernel)"""
            flop = sum(t[0] for t in entries)
            ms_ = sum(t[1].elapsed_time(t[2]) for t in entries)
            alg = flop / (ms_ * 1e-3) / 1e12
            issued = sum(t[0] * t[3] for t in entries) / (ms_ * 1e-3) / 1e12
            # (a mix of pipes - only when f32-MFMA launches sit beside split ones - is priced launch by launch)
            pipe_time = sum(t[0] * t[3] / (pipe_peak_of(t[3]) * 1e12) for t in entries)
            ceiling = flop / pipe_time / 1e12                 # fp32-equivalent rate with the pipe at its dense peak throughout
            peak = pipe_peak_of(max(t[3] for t in entries))
            return {"achieved": round(alg, 2), "peak": peak, "frac": round(alg / peak, 4),
                    "mfma_issue_tflops": round(issued, 1), "mfma_issue_frac": round(issued / peak, 4),
                    "emulation_ceiling_tflops": round(ceiling, 1), "emulation_ceiling_frac": round(alg / ceiling, 4),
                    "ms_per_step": round(ms_ / a.steps, 4), "launches_per_step": round(len(entries) / a.steps, 2),
                    "avg_launch_us": round(1e3 * ms_ / len(entries), 2), "gflop_per_launch": round(flop / len(entries) / 1e9, 3)}

        by_kernel = {}
        for t in timing:
            by_kernel.setdefault(t[4], []).append(t)
        table = {k: rates(v) for k, v in by_kernel.items()}
        dominant = max(table, key=lambda k: table[k]["ms_per_step"])
        dom, fam = table[dominant], rates(timing)
        gemm_bytes = kernels.GEMM_BYTES
        ms = sum(t[1].elapsed_time(t[2]) for t in timing)
        roofline = {"bound": "mfma", "kernel": dominant, "unit": "TFLOP/s",
                    "achieved": dom["achieved"], "peak": dom["peak"], "frac": dom["frac"],
                    "mfma_issue_frac": dom["mfma_issue_frac"], "emulation_ceiling_frac": dom["emulation_ceiling_frac"],
                    "avg_launch_us": dom["avg_launch_us"], "launches_per_step": dom["launches_per_step"],
                    "gflop_per_launch": dom["gflop_per_launch"],
                    "traffic": traffic, "traffic_source": traffic_note, "traffic_unit": "HBM bytes per GEMM launch (family average)",
                    "frac_is": "achieved / peak with achieved = ALGORITHMIC fp32 FLOP (2 M N K of the dominant kernel's launches) per "
                               "second of its measured launch time (HIP events on the launch stream) and peak = the dense peak of the "
                               "matrix pipe it runs on (MI355X_MICROARCH.md: 2500 TF/s f16 / bf16, 157.3 f32).  The f16x2 arithmetic "
                               "spends 3 matrix-pipe products per fp32 product: mfma_issue_frac = 3 x frac is how busy the pipe is, "
                               "emulation_ceiling_frac = frac against peak / 3, the most a 3-product emulation can reach.  (Rounds 3-4 "
                               "printed mfma_issue_frac of the whole family as `frac`.)",
                    "gemm_family": {**fam, "share_of_step_time": round(ms / (1e3 * dt_inst), 3),
                                    "algorithmic_bytes_per_launch": round(sum(gemm_bytes) / max(len(gemm_bytes), 1)),
                                    "gflop_per_step": round(sum(t[0] for t in timing) / a.steps / 1e9, 1)},
                    "kernels": table,
                    "measured_in": f"a pass of its own: the same {a.steps} steps (batches resident) with two HIP events around every GEMM "
                                   f"call and the weight-gradient products on the main stream ({round(1e3 * dt_inst / a.steps, 3)} ms/step in "
                                   f"that pass; `value` comes from the un-instrumented passes)"}

    dtype = {"f32": "f32",
             "f16x2": "f32 (GEMM and attention operands scaled by powers of two and split into 2 f16 terms on the f16 MFMA pipe, "
                      "f32 accumulate; the rest f32)",
             "auto": "f32 (GEMM and attention operands scaled by powers of two and split into 2 f16 terms [activation x weight "
                     "products, weight-gradient products whose operands come with a uniform scale, attention with head size "
                     "64 / 32] or exactly into 3 bf16 terms [the other weight-gradient products; everything when tokens x "
                     "d_model < 2^20] on the f16 / bf16 MFMA pipe, f32 accumulate; the rest f32)"}.get(
        a.gemm_mode, "f32 (GEMM operands split exactly into 3 bf16 terms on the bf16 MFMA pipe, f32 accumulate; the rest f32)")
    if rank == 0:
        shape = (f"{a.batch} proteins x L={a.length} per GPU" if not a.ragged else
                 f"ragged batches, L <= {a.length}" + (f", binned-random batching with a budget of {a.batch * a.length} residues "
                                                       f"per GPU and batch" if a.ragged == "binned" else f", {a.batch} proteins per GPU"))
        out = {
            "metric": f"train-step residues/sec ({a.model.split('|')[0]} d{a.d_model}, {a.loss} loss)",
            "value": round(world * n_res_timed / dt, 1), "unit": "residues/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"{a.model} d_model={a.d_model} n_layers={a.n_layers} n_head={a.n_head} "
                                   f"d_ff={a.d_ff}, -l {a.loss}, {a.optimizer} lr 1e-4 wd 0.01 clip 1, dropout {a.dropout}, "
                                   f"{shape} (BASELINE.json configs[{a.config - 1}])",
                       "global_batch": a.batch * world, "seq_len": a.length, "parallelism": f"dp{world}",
                       "residues_per_step": round(world * n_res_timed / a.steps, 1),
                       "last_loss": {k: float(losses[k]) for k in ("drmsd-full", "lndrmsd-full", "mse-full")}},
            "passes": {"ms_per_step": [round(1e3 * d / a.steps, 3) for d in passes],
                       "spread_rel": round((max(passes) - min(passes)) / dt, 4),
                       "what": f"{len(passes)} timed passes of {a.steps} steps each, back to back; value / ms_per_step are the median pass; "
                               f"every step's batch is uploaded from pinned host memory inside the step (dataset.DevicePrefetcher)"},
            "resident": {"value": round(world * n_res_timed / dt_res, 1), "ms_per_step": round(1e3 * dt_res / a.steps, 3),
                         "what": "one more pass with the batches already resident in HBM (the `value` of rounds 1-4)"},
            "strong_scaling": strong,
            "arithmetic_modes": {a.gemm_mode: {"ms_per_step": round(1e3 * dt_res / a.steps, 3)}, **sweep,
                                 "what": "the same steps with the batches resident, per arithmetic (compare with `resident`)"},
            "roofline": roofline,
        }
        # the guard of AUTO's bound-derived f16x2 scales (models/encoder_only.py AutoGuard): products per step that left their
        # bound for exact scales / bf16x3 because the measured slack of the bound exceeded 8 binades, and what was measured
        guard = model.auto_guard.report()
        out["auto_fallbacks_per_step"] = round(guard["fallbacks_per_step"], 3)
        # layers x passes whose attention forward kernel handed its dropout decisions to the fused backward kernel
        out["attn_keep_bits_layer_passes"] = int(model.__dict__.get("_attn_bits_passes", 0))
        # ... and whose FFN layer 1 left the 1-bit gate of its output for the gated dX product of layer 2
        out["ffn_gate_mask_layer_passes"] = int(model.__dict__.get("_gate_mask_passes", 0))
        # forward passes that had to prepare the weights' scales / bounds / planes themselves (the others found them left behind
        # by the optimizer step: csrc/wprep.hip)
        out["kv_plane_layer_passes"] = int(model.__dict__.get("_kv_plane_passes", 0))
        out["weights_prep_launches"] = int(model.__dict__.get("_prep_launches", 0))
        out["auto_guard"] = guard
        out["communication"] = comm
        if world == 1 and not a.no_cpu_baseline:
            keys = ("seq", "true_ang", "true_crd")
            pick = min(range(nb), key=lambda i: abs(host_batches[i][0].shape[1] - 200)) if a.ragged == "binned" else 0
            cpu_batch = dict(zip(keys, (t.clone() for t in host_batches[pick])))
            params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}   # the same model, reference keys
            oracle_step = None
            try:
                oracle_step, out["cpu_baseline"] = cpu_baseline(a, cpu_batch, params)
            except Exception as e:                       # the bench line must not depend on the checker's health
                out["cpu_baseline"] = {"value": None, "unit": "residues/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
            # "dRMSD-loss delta vs ref" (the metric's second half): the HIP step against the oracle step that was just timed
            if oracle_step is not None and a.optimizer == "sgd":
                try:
                    out["parity"] = parity_block(a, model, opt, args, cpu_batch, params, oracle_step, dev)
                except Exception as e:
                    out["parity"] = {"ok": False, "failed": f"{type(e).__name__}: {e}"}
            else:
                out["parity"] = None
        else:
            out["cpu_baseline"] = None
            out["parity"] = None
        print(json.dumps(out), flush=True)
    dp.shutdown()


if __name__ == "__main__":
    main()
