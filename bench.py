#!/usr/bin/env python
"""Headline benchmark: train-step residues/s of the enc-only d512 model with the dRMSD loss on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[3], the one its metric is quoted on): `-m enc-only -dm 512 -nl 6 -nh 8
-dih 2048 -l drmsd`, SGD lr 1e-4 wd 0.01 clip 1, dropout 0.1 ON, 32 synthetic proteins of L=512 per GPU
(weak scaling: the global batch is 32 x N).  One step = zero_grad, forward, NeRF + dRMSD loss + backward,
gradient all-reduce, clip, optimizer step (train.train_step = reference train.py:36-46), inputs already
resident in HBM.  Prints ONE JSON line on rank 0 (contract in the task description) with
  roofline     - the fp32 MFMA GEMM kernel (dominant: ~80% of the step), timed live with HIP events on
                 its launch stream during the timed steps: algorithmic FLOP / measured kernel time
                 against the 157.3 TF/s dense f32 matrix peak;
  cpu_baseline - the CPU oracle (a port of the reference's --no_cuda path) on a bounded sample of the
                 same workload, on the host cores of this box.
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
BF16_MFMA_PEAK_TFLOPS = 2500.0        # same table: dense bf16 matrix peak (the marketing figure includes 2:1 sparsity)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="proteins per GPU")
    ap.add_argument("--length", type=int, default=512)
    ap.add_argument("--d_model", type=int, default=512)
    ap.add_argument("--n_layers", type=int, default=6)
    ap.add_argument("--n_head", type=int, default=8)
    ap.add_argument("--d_ff", type=int, default=2048)
    ap.add_argument("--loss", default="drmsd")
    ap.add_argument("--optimizer", default="sgd")
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-proteins", type=int, default=2, help="proteins in the bounded CPU-baseline sample")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--gemm-mode", default="auto", choices=["f32", "bf16x3", "bf16x3full", "f16x2", "auto"],
                    help="arithmetic of the encoder GEMMs (include/ptamd.h: ptamd_gemm_set_mode)")
    return ap.parse_args()


def cpu_baseline(a, batch_cpu, angle_means):
    """Time ONE step of the CPU oracle on `--cpu-proteins` proteins of the same workload (1 thread, sequential
    loss, like the reference with --sequential_drmsd_loss; the reference pins torch.set_num_threads(1), train.py:344)."""
    from oracle import encoder as oenc, step as ostep
    n = a.cpu_proteins
    torch.set_num_threads(1)
    params = oenc.init_params(a.n_layers, a.d_model, a.d_ff, a.length, angle_means, seed=11731)
    trainer = ostep.CpuTrainer(params, a.n_head, loss=a.loss, optimizer=a.optimizer, lr=1e-4, clip=1.0)
    seq, ang, crd = (batch_cpu[k][:n] for k in ("seq", "true_ang", "true_crd"))
    res_per_s, dt = ostep.time_cpu_steps(trainer, (seq, ang, crd), n_steps=1)
    return {"value": round(res_per_s, 2), "unit": "residues/s", "cores": 1, "kind": "port",
            "sample": f"1 step of oracle.step.CpuTrainer on {n} of the {a.batch} proteins (L={a.length}, same model, "
                      f"dropout 0, sequential loss, torch threads=1): {dt:.1f} s"}


def main():
    a = parse()
    from protein_transformer_amd import dp, kernels, synthetic
    from protein_transformer_amd.models.encoder_only import EncoderOnlyTransformer
    from protein_transformer_amd.optim import FusedAdam, FusedSGD
    from protein_transformer_amd.protein.Sequence import VOCAB
    from protein_transformer_amd.protein.Structure import nerf_forward
    from protein_transformer_amd.train import train_step

    dp.init_from_env()
    world, rank = dp.world_size(), dp.rank()
    if world != a.gpus:
        sys.exit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    dev = torch.device("cuda", dp.local_rank())
    torch.cuda.set_device(dev)
    kernels.set_gemm_mode({"f32": kernels.GEMM_F32, "bf16x3": kernels.GEMM_BF16X3, "bf16x3full": kernels.GEMM_BF16X3_FULL,
                           "f16x2": kernels.GEMM_F16X2, "auto": kernels.GEMM_AUTO}[a.gemm_mode])

    # ---- synthetic, device-resident batches (two per rank, alternated)
    build = lambda ang, seq: nerf_forward(ang.to(dev), seq.to(dev))[0]            # noqa: E731
    batches_cpu = [synthetic.make_batch([a.length] * a.batch, seed=synthetic.DEFAULT_SEED + 97 * rank + i,
                                        build_coords=build) for i in range(2)]
    angle_means = synthetic.angle_means(batches_cpu[0]["true_ang"])
    batches = [tuple(b[k].to(dev) for k in ("seq", "true_ang", "true_crd")) for b in batches_cpu]
    n_res = int((batches[0][0] != 20).sum())

    torch.manual_seed(synthetic.DEFAULT_SEED)
    model = EncoderOnlyTransformer(a.n_layers, a.n_head, a.d_model, a.d_ff, a.length, VOCAB, angle_means, True,
                                   dropout=a.dropout).to(dev).train()
    model.dropout_seed += 7919 * rank
    dp.attach(model)
    opt = (FusedAdam(model, lr=1e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=10e-3) if a.optimizer == "adam"
           else FusedSGD(model, lr=1e-4, weight_decay=10e-3))
    args = types.SimpleNamespace(loss=a.loss, combined_drmsd_weight=0.5, backbone_loss=False, clip=1.0)

    def step(i):
        return train_step(model, opt, args, *batches[i % 2])

    for i in range(a.warmup):
        losses = step(i)
    timing = None if a.no_kernel_timing else []
    if timing is not None:      # GEMM launches per step are counted in the warm-up; events are created up front
        kernels.GEMM_TIMING, kernels.GEMM_EVENT_POOL = [], [torch.cuda.Event(enable_timing=True) for _ in range(400)]
        step(0)
        per_step = len(kernels.GEMM_TIMING)
        kernels.GEMM_TIMING = None
        kernels.GEMM_BYTES.clear()
        kernels.GEMM_EVENT_POOL = [torch.cuda.Event(enable_timing=True) for _ in range(2 * per_step * a.steps + 8)]
        for e in kernels.GEMM_EVENT_POOL:
            e.record()          # materialise the underlying hipEvents outside the timed region
    gc.collect()
    gc.disable()                # a generation-2 collection over the event pool costs ~60 ms when it lands in the timed steps
    dp.barrier()
    torch.cuda.synchronize()
    kernels.GEMM_TIMING = timing
    t0 = time.perf_counter()
    for i in range(a.steps):
        losses = step(i)
    dp.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    kernels.GEMM_TIMING = None
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())

    traffic = None          # HBM bytes per GEMM launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    tpath = os.path.join(ROOT, "profiles", "r01_gemm_hbm_traffic.json")
    if os.path.exists(tpath) and (a.batch, a.length, a.d_model, a.n_layers) == (32, 512, 512, 6):
        with open(tpath) as f:
            traffic = round(json.load(f)["hbm_bytes_per_launch"])
    roofline = None
    if timing:
        flops = sum(t[0] for t in timing)
        gemm_bytes = kernels.GEMM_BYTES
        ms = sum(t[1].elapsed_time(t[2]) for t in timing)
        achieved = flops / (ms * 1e-3) / 1e12
        # Every launch runs one fp32 product as `products` matrix-pipe products (1: f32 MFMA; 3: two f16 terms; 6 / 9:
        # three bf16 terms).  The f32-equivalent ceiling of the launch mix is the rate at which the mix would run with
        # the matrix pipe at its dense peak throughout: sum(flop) / sum(flop_i * products_i / pipe peak_i).
        pipe_time = sum(t[0] * t[3] / ((F32_MFMA_PEAK_TFLOPS if t[3] == 1 else BF16_MFMA_PEAK_TFLOPS) * 1e12) for t in timing)
        peak = flops / pipe_time / 1e12
        issued = sum(t[0] * t[3] for t in timing) / (ms * 1e-3) / 1e12
        by_products = {}
        for t in timing:
            e = by_products.setdefault(t[3], [0, 0.0, 0.0])
            e[0] += 1; e[1] += t[0]; e[2] += t[1].elapsed_time(t[2])
        names = {1: "gemm_f32_mfma_kernel (v_mfma_f32_32x32x2_f32)",
                 3: "gemm_bf16x3_mfma_kernel<NPROD=3> (two row-scaled f16 terms, 3 x v_mfma_f32_32x32x16_f16 per fp32 product; "
                    "time includes gemm_row_scale_kernel)",
                 6: "gemm_bf16x3_mfma_kernel<NPROD=6> (three bf16 terms, 6 x v_mfma_f32_32x32x16_bf16 per fp32 product)",
                 9: "gemm_bf16x3_mfma_kernel<NPROD=9> (three bf16 terms, all 9 products)"}
        mix = {names[k]: {"launches_per_step": v[0] // a.steps, "tflops_f32_equivalent": round(v[1] / (v[2] * 1e-3) / 1e12, 1),
                          "peak_f32_equivalent": round((F32_MFMA_PEAK_TFLOPS if k == 1 else BF16_MFMA_PEAK_TFLOPS) / k, 1),
                          "ms_per_step": round(v[2] / a.steps, 3)} for k, v in sorted(by_products.items())}
        kern = "ptamd_gemm: " + " + ".join(f"{v[0] // a.steps} x NPROD={k}" for k, v in sorted(by_products.items()))
        products = issued / achieved
        roofline = {"bound": "mfma", "kernel": kern,
                    "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic,
                    "achieved_is": "algorithmic fp32 FLOP (2*M*N*K) per second of GEMM kernel time",
                    "peak_is": "f32-equivalent ceiling of the launch mix: sum(flop) / sum(flop_i * products_i / dense MFMA peak_i)",
                    "mfma_flops_issued_tflops": round(issued, 1),
                    "mfma_instruction_peak_tflops": BF16_MFMA_PEAK_TFLOPS if products > 1 else F32_MFMA_PEAK_TFLOPS,
                    "launch_mix": mix,
                    "traffic_unit": "HBM bytes per launch (PMC, profiles/r01_gemm_hbm_traffic.json)",
                    "algorithmic_bytes_per_launch": round(sum(b for b in gemm_bytes) / max(len(gemm_bytes), 1)),
                    "launches_per_step": len(timing) // a.steps, "avg_launch_us": round(1e3 * ms / len(timing), 2),
                    "gflop_per_step": round(flops / a.steps / 1e9, 1),
                    "share_of_step_time": round(ms / (dt * 1e3), 3)}

    dtype = {kernels.GEMM_F32: "f32",
             kernels.GEMM_F16X2: "f32 (GEMM operands row-scaled and split into 2 f16 terms on the f16 MFMA pipe, f32 accumulate; "
                                 "attention operands split into 3 bf16 terms; the rest f32)",
             kernels.GEMM_AUTO: "f32 (GEMM operands split into 2 row-scaled f16 terms [activation x weight products] or exactly "
                                "into 3 bf16 terms [weight-gradient products, attention] on the f16 / bf16 MFMA pipe, f32 "
                                "accumulate; the rest f32)"}.get(
        kernels.get_gemm_mode(),
        "f32 (GEMM operands split exactly into 3 bf16 terms on the bf16 MFMA pipe, f32 accumulate; the rest f32)")
    # the stored line of the same bench in the exact-f32 MFMA mode (python bench.py --gemm-mode f32), for comparison
    f32_ref = None
    fpath = os.path.join(ROOT, "profiles", "r01_v9_bench_gemm_mode_f32.json")
    if roofline is not None and kernels.get_gemm_mode() != kernels.GEMM_F32 and os.path.exists(fpath):
        with open(fpath) as f:
            r = json.load(f)
        f32_ref = {"ms_per_step": r["ms_per_step"], "residues_per_s": r["value"], "gemm_tflops": r["roofline"]["achieved"],
                   "gemm_frac_of_f32_mfma_peak": r["roofline"]["frac"], "source": "profiles/r01_v9_bench_gemm_mode_f32.json"}
    if roofline is not None:
        roofline["exact_f32_mfma_mode_reference"] = f32_ref
    if rank == 0:
        out = {
            "metric": "train-step residues/sec (enc-only d512, dRMSD loss)",
            "value": round(world * n_res * a.steps / dt, 1), "unit": "residues/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"enc-only d_model={a.d_model} n_layers={a.n_layers} n_head={a.n_head} "
                                   f"d_ff={a.d_ff}, -l {a.loss}, {a.optimizer} lr 1e-4 wd 0.01 clip 1, dropout {a.dropout}, "
                                   f"{a.batch} proteins x L={a.length} per GPU (BASELINE.json configs[3])",
                       "global_batch": a.batch * world, "seq_len": a.length, "parallelism": f"dp{world}",
                       "last_loss": {k: float(losses[k]) for k in ("drmsd-full", "lndrmsd-full")}},
            "roofline": roofline,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a, batches_cpu[0], angle_means)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    dp.shutdown()


if __name__ == "__main__":
    main()
