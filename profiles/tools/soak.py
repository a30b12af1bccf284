#!/usr/bin/env python3
"""Soak: N training steps of the headline configuration through the product's own data path (DevicePrefetcher, packed batches),
reporting every N / 10 steps the step time, the loss, the allocator's reserved / allocated bytes and the guard's state - memory
must be flat, the loss finite and falling, no site off its bound.  python profiles/tools/soak.py [--steps 3000]"""
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
    from protein_transformer_amd.dataset import DevicePrefetcher
    from protein_transformer_amd.optim import FusedSGD
    from protein_transformer_amd.train import train_step
    dp.init_from_env()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    host, angle_means, _ = bench.make_batches(a, 0, dev, 2)
    torch.manual_seed(synthetic.DEFAULT_SEED)
    model = bench.make_model(a, angle_means, dev)
    model.gemm_mode = kernels.GEMM_AUTO
    dp.attach(model)
    opt = FusedSGD(model, lr=1e-4, weight_decay=10e-3)
    opt.zero_grad_in_step = True          # as train.setup_model_optimizer_scheduler sets it for the product's loop
    args = types.SimpleNamespace(loss=a.loss, combined_drmsd_weight=0.5, backbone_loss=False, clip=1.0)
    N = a.steps
    chunk = max(1, N // 10)
    done = 0
    gc_every = int(os.environ.get("SOAK_GC_EVERY", "0"))      # experiment: a full collection every so many steps
    if os.environ.get("SOAK_GC_OFF"):
        import gc
        gc.disable()
    while done < N:
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        for k, (seq, ang, crd, n) in enumerate(DevicePrefetcher((host[i % 2] for i in range(chunk)), dev)):
            out = train_step(model, opt, args, seq, ang, crd, n_res=n)
            if gc_every and (k + 1) % gc_every == 0:
                import gc
                gc.collect()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / chunk * 1e3
        done += chunk
        flat, g = model.flat_parameters()
        g_ = model.auto_guard
        st = torch.cuda.memory_stats()
        print(f"step {done:6d}  {dt:7.3f} ms/step  loss {float(out['loss']):.5f}  finite {bool(torch.isfinite(flat).all())}  "
              f"allocated {torch.cuda.memory_allocated() / 2**20:8.1f} MiB (peak {torch.cuda.max_memory_allocated() / 2**20:8.1f})  reserved {torch.cuda.memory_reserved() / 2**20:8.1f} MiB  "
              f"segments {st.get('segment.all.current', 0)} allocs/retries {st.get('num_alloc_retries', 0)}  "
              f"guard: measured {g_.measured_steps} off {int(g_.off.sum())} wide {int(g_.wide.sum())}", flush=True)


if __name__ == "__main__":
    main()
