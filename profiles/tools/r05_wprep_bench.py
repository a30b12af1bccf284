#!/usr/bin/env python
"""Round 5: the weights of a step in one pass (csrc/wprep.hip) against the launches it replaces, config-4 model, in isolation.
    python profiles/tools/r05_wprep_bench.py  ->  microseconds per call (HIP events, 100 repetitions each)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K  # noqa: E402
from protein_transformer_amd import synthetic  # noqa: E402
from protein_transformer_amd.models.encoder_only import EncoderOnlyTransformer  # noqa: E402
from protein_transformer_amd.protein.Sequence import VOCAB  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
am = synthetic.angle_means(synthetic.make_batch([64], seed=1)["true_ang"])
m = EncoderOnlyTransformer(6, 8, 512, 2048, 512, VOCAB, am, True, dropout=0.1).to(dev).train()
flat, grad = m.flat_parameters()
grad.normal_(0, 1e-3)
m.__dict__["_fwd_grad"] = True
sq = torch.ones(1, device=dev)


def timeit(fn, reps=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


m.weights_prep = False
m._step_scales(flat, K.GEMM_AUTO, 0.1, 0.1, hp=True)
cache = next(iter(m.__dict__["_scale_caches"].values()))
prep = cache["prep"]


def old_path():
    K.sgd_step(flat, grad, sq, 1.0, 0.0, 0.0)
    m._step_scales(flat, K.GEMM_AUTO, 0.1, 0.1, hp=True)


print("sgd_step alone                                   %7.1f us" % timeit(lambda: K.sgd_step(flat, grad, sq, 1.0, 0.0, 0.0)))
print("sgd_step + scales + bounds + split_rows + _cols  %7.1f us" % timeit(old_path))
print("scales + bounds + split_rows + _cols             %7.1f us" % timeit(lambda: m._step_scales(flat, K.GEMM_AUTO, 0.1, 0.1, hp=True)))
print("ptamd_weights_prep (no update, planes)           %7.1f us" % timeit(lambda: prep.prepare(flat, with_planes=True)))
print("ptamd_weights_prep (no update, no planes)        %7.1f us" % timeit(lambda: prep.prepare(flat, with_planes=False)))
print("ptamd_sgd_step_prep (planes)                     %7.1f us" % timeit(lambda: prep.sgd_step(flat, grad, sq, 1.0, 0.0, 0.0, with_planes=True)))
print("ptamd_sgd_step_prep (no planes)                  %7.1f us" % timeit(lambda: prep.sgd_step(flat, grad, sq, 1.0, 0.0, 0.0, with_planes=False)))
mm, vv = torch.zeros_like(flat), torch.zeros_like(flat)
print("adam_step alone                                  %7.1f us" % timeit(lambda: K.adam_step(flat, grad, mm, vv, sq, 1.0, 0.0, 0.9, 0.98, 1e-9, 0.0, 3)))
print("ptamd_adam_step_prep (planes)                    %7.1f us" % timeit(lambda: prep.adam_step(flat, grad, mm, vv, sq, 1.0, 0.0, 0.9, 0.98, 1e-9, 0.0, 3, with_planes=True)))
