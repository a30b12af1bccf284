"""Accuracy of the attention kernels in the three arithmetics against dense fp64 attention, on ordinary data and on
operands with a wide range of row magnitudes (the scaling groups of the f16x2 kernels).
python profiles/tools/r02_attn_accuracy.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_transformer_amd import kernels as K   # noqa: E402

dev = torch.device("cuda:0")


def ref(qkv, key_ok, H, dout):
    B, L, D3 = qkv.shape
    D = D3 // 3
    dk = D // H
    qkv = qkv.clone().requires_grad_()
    q, k, v = (t.reshape(B, L, H, dk).transpose(1, 2) for t in qkv.split(D, dim=-1))
    s = q @ k.transpose(-2, -1) / np.sqrt(dk)
    s = s.masked_fill(~key_ok[:, None, None, :], -np.inf)
    o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, L, D)
    o.backward(dout)
    return o.detach(), qkv.grad


def rel(a, b):
    return ((a.double().cpu() - b).norm() / b.norm()).item(), ((a.double().cpu() - b).abs().max() / b.abs().max()).item()


for name, wide in (("ordinary", False), ("wide row ranges", True)):
    for B, L, H, dk in [(2, 512, 8, 64), (3, 300, 4, 32)]:
        g = torch.Generator().manual_seed(11)
        D = H * dk
        qkv = torch.randn(B, L, 3 * D, generator=g, dtype=torch.float64)
        dout = torch.randn(B, L, D, generator=g, dtype=torch.float64)
        if wide:
            qkv[:, :, D:2 * D] *= 10 ** (torch.rand(B, L, 1, generator=g, dtype=torch.float64) * 2.5 - 2)      # K rows 1e-2 .. 3
            qkv[:, :, 2 * D:] *= 10 ** (torch.rand(B, L, 1, generator=g, dtype=torch.float64) * 4 - 3)         # V rows 1e-3 .. 10
            dout *= 10 ** (torch.rand(B, L, 1, generator=g, dtype=torch.float64) * 4 - 8)                      # dO rows 1e-8 .. 1e-4
        qkv = qkv.float().double()
        dout = dout.float().double()
        seq = torch.randint(0, 20, (B, L), generator=g)
        seq[-1, L - 37:] = 20
        o_ref, g_ref = ref(qkv, seq != 20, H, dout)
        qd = qkv.float().view(B * L, 3 * D).to(dev)
        for mode, mname in ((K.GEMM_F32, "f32"), (K.GEMM_BF16X3, "bf16x3"), (K.GEMM_F16X2, "f16x2")):
            o, lse = K.attention_fwd(qd, seq.to(dev), H, 0.0, 0, 0, arith=mode)
            dq = K.attention_bwd(qd, seq.to(dev), o, dout.float().view(B * L, D).to(dev), lse, H, 0.0, 0, 0, arith=mode)
            eo, eg = rel(o.view(B, L, D), o_ref), rel(dq.view(B, L, 3 * D), g_ref)
            parts = [rel(dq.view(B, L, 3 * D)[:, :, i * D:(i + 1) * D], g_ref[:, :, i * D:(i + 1) * D])[0] for i in range(3)]
            print(f"{name:16s} dk={dk} L={L} {mname:7s} out relL2 {eo[0]:.2e} max/max {eo[1]:.2e} | dqkv relL2 {eg[0]:.2e} max/max {eg[1]:.2e}"
                  f" | dQ {parts[0]:.2e} dK {parts[1]:.2e} dV {parts[2]:.2e}  finite={bool(torch.isfinite(dq).all())}")
