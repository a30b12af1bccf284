#!/usr/bin/env python3
"""Host-side profile of the training step (cProfile over N steps of bench.py's workload): where the Python time of a
launch-bound step goes.  python profiles/tools/host_profile.py --batch 4 --steps 200"""
import cProfile
import os
import pstats
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def main():
    a = bench.parse()
    from protein_transformer_amd import dp, kernels, synthetic
    from protein_transformer_amd.optim import FusedSGD
    from protein_transformer_amd.train import train_step
    dp.init_from_env()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    host, angle_means, _ = bench.make_batches(a, 0, dev, 2)
    res = [tuple(t.to(dev) for t in b) for b in host]
    torch.manual_seed(synthetic.DEFAULT_SEED)
    model = bench.make_model(a, angle_means, dev)
    model.gemm_mode = kernels.GEMM_AUTO
    opt = FusedSGD(model, lr=1e-4, weight_decay=10e-3)
    opt.zero_grad_in_step = True
    args = types.SimpleNamespace(loss=a.loss, combined_drmsd_weight=0.5, backbone_loss=False, clip=1.0)
    n_res = int((res[0][0] != 20).sum())
    for i in range(20):
        train_step(model, opt, args, *res[i % 2], n_res=n_res)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        train_step(model, opt, args, *res[i % 2], n_res=n_res)
    torch.cuda.synchronize()
    print(f"{a.batch} proteins: {1e3 * (time.perf_counter() - t0) / a.steps:.3f} ms/step wall (resident batches)")
    # the loop's own time against the GPU's: every step waits for its own loss statistics (LossReport.wait, behind the
    # enqueued backward pass), so the host never runs ahead by more than a backward pass - where the GPU is the limit the
    # loop takes the GPU's time and the GPU finishes a fraction of a step later; where the loop takes LONGER than the
    # GPU-bound step of a fast-host box (2.64 ms at 4 proteins), the host is the limit
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40):
            train_step(model, opt, args, *res[i % 2], n_res=n_res)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"   40 steps: the loop took {1e3 * (t1 - t0) / 40:.3f} ms/step; the GPU finished {1e3 * (t2 - t1):.1f} ms "
              f"after the loop ({1e3 * (t2 - t0) / 40:.3f} ms/step in all)")
    # the backward pass runs on autograd's device thread: a profiler of its own around _EncoderFn.backward
    from protein_transformer_amd.models import encoder_only as EO
    pb = cProfile.Profile()
    real_bwd = EO._EncoderFn.backward

    def bwd(ctx, dpred):
        pb.enable()
        try:
            return real_bwd(ctx, dpred)
        finally:
            pb.disable()
    EO._EncoderFn.backward = staticmethod(bwd)
    pr = cProfile.Profile()
    pr.enable()
    for i in range(a.steps):
        train_step(model, opt, args, *res[i % 2], n_res=n_res)
    pr.disable()
    torch.cuda.synchronize()
    print("---- main thread (forward, loss path, optimizer; run_backward = waiting for the backward thread)")
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
    print("---- backward thread (_EncoderFn.backward)")
    pstats.Stats(pb).sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
