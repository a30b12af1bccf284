#!/usr/bin/env python3
"""What does the batch upload inside the step cost, and which part of it?  (bench.py: `value` 10.28 ms against `resident`
10.18 at 32 proteins, 3.22 against 3.05 at 4.)  Same model, same batches, K steps per variant, alternating:
  resident        - batches already in HBM
  prefetcher      - dataset.DevicePrefetcher (three copies on a side stream, one batch ahead, record_stream)
  same-stream     - three copies on the compute stream at the top of the step
  packed          - ONE copy of a pre-packed pinned buffer on the side stream, one batch ahead (views on the device)
  packed-same     - one copy on the compute stream
usage: python profiles/tools/r05_upload_cost.py [--batch B] [--steps K]"""
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
    resident = [tuple(t.to(dev) for t in b) for b in host]
    res_of = [int((b[0] != 20).sum()) for b in host]
    torch.manual_seed(synthetic.DEFAULT_SEED)
    model = bench.make_model(a, angle_means, dev)
    model.gemm_mode = kernels.GEMM_AUTO
    dp.attach(model)
    opt = FusedSGD(model, lr=1e-4, weight_decay=10e-3)
    args = types.SimpleNamespace(loss=a.loss, combined_drmsd_weight=0.5, backbone_loss=False, clip=1.0)
    K = a.steps

    # packed form: [seq int64 | ang f32 | crd f32] in one pinned byte buffer per batch
    packed = []
    for s, g, c in host:
        parts = [s.contiguous().view(-1).view(torch.uint8), g.contiguous().view(-1).view(torch.uint8), c.contiguous().view(-1).view(torch.uint8)]
        buf = torch.cat(parts).pin_memory()
        packed.append((buf, s.shape, g.shape, c.shape, [p.numel() for p in parts]))

    def unpack(d, shapes):
        _, ss, gs, cs, n = shapes
        return (d[:n[0]].view(torch.int64).view(ss), d[n[0]:n[0] + n[1]].view(torch.float32).view(gs),
                d[n[0] + n[1]:].view(torch.float32).view(cs))

    def run_resident():
        for i in range(K):
            train_step(model, opt, args, *resident[i % 2], n_res=res_of[i % 2])

    def run_prefetcher():
        for seq, ang, crd, n in DevicePrefetcher((host[i % 2] for i in range(K)), dev):
            train_step(model, opt, args, seq, ang, crd, n_res=n)

    def run_same():
        for i in range(K):
            seq, ang, crd = (t.to(dev, non_blocking=True) for t in host[i % 2])
            train_step(model, opt, args, seq, ang, crd, n_res=res_of[i % 2])

    side = torch.cuda.Stream(dev)

    def run_packed():
        cur = torch.cuda.current_stream(dev)

        def fetch(i):
            with torch.cuda.stream(side):
                d = packed[i % 2][0].to(dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
            return d, ev
        nxt = fetch(0)
        for i in range(K):
            d, ev = nxt
            cur.wait_event(ev)
            d.record_stream(cur)
            if i + 1 < K:
                nxt = fetch(i + 1)
            seq, ang, crd = unpack(d, packed[i % 2])
            train_step(model, opt, args, seq, ang, crd, n_res=res_of[i % 2])

    def run_packed_same():
        for i in range(K):
            d = packed[i % 2][0].to(dev, non_blocking=True)
            seq, ang, crd = unpack(d, packed[i % 2])
            train_step(model, opt, args, seq, ang, crd, n_res=res_of[i % 2])

    variants = [("resident", run_resident), ("prefetcher", run_prefetcher), ("same-stream", run_same), ("packed", run_packed),
                ("packed-same", run_packed_same)]
    for _ in range(a.warmup):
        train_step(model, opt, args, *resident[0], n_res=res_of[0])
    res = {k: [] for k, _ in variants}
    for rep in range(4):
        for name, fn in variants:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            res[name].append(1e3 * (time.perf_counter() - t0) / K)
    print(f"batch {a.batch} x {a.length}, {K} steps per variant, ms/step of 4 alternating repetitions")
    for name, _ in variants:
        v = res[name]
        print(f"  {name:12s} " + "  ".join(f"{x:7.3f}" for x in v) + f"   median of the last three {sorted(v[1:])[1]:7.3f}")


if __name__ == "__main__":
    main()
