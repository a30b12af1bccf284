#!/usr/bin/env python3
"""Per-step times of the headline configuration (HIP events around every step, resident batches): where do the slow passes of
the bench come from - the start of a timed region, or the guard's measuring steps?  python profiles/tools/r05_step_times.py"""
import os
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
    resident = [tuple(t.to(dev) for t in b) for b in host]
    res_of = [int((b[0] != 20).sum()) for b in host]
    torch.manual_seed(synthetic.DEFAULT_SEED)
    model = bench.make_model(a, angle_means, dev)
    model.gemm_mode = kernels.GEMM_AUTO
    dp.attach(model)
    opt = FusedSGD(model, lr=1e-4, weight_decay=10e-3)
    args = types.SimpleNamespace(loss=a.loss, combined_drmsd_weight=0.5, backbone_loss=False, clip=1.0)
    N = 200
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    for e in ev:
        e.record()
    torch.cuda.synchronize()
    for pause in (0.0, 0.5):
        time.sleep(pause)
        ev[0].record()
        for i in range(N):
            train_step(model, opt, args, *resident[i % 2], n_res=res_of[i % 2])
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
        print(f"after a host pause of {pause} s: {N} steps, mean {sum(ms) / N:.3f} ms; by tens: " +
              " ".join(f"{sum(ms[j:j + 10]) / 10:.2f}" for j in range(0, N, 10)))
        srt = sorted(ms)
        print("  median", f"{srt[N // 2]:.3f}", " steps over median + 0.5 ms:", [(i, round(m, 2)) for i, m in enumerate(ms) if m > srt[N // 2] + 0.5])


if __name__ == "__main__":
    main()
