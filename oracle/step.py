"""Oracle: loss dispatch and one full training step on the CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates /root/reference/protein_transformer/train.py:
  train_epoch 28-54 (zero_grad, forward, get_losses, clip, optimizer.step),
  get_losses 57-111, setup_model_optimizer_scheduler 368-393 (SGD lr wd=0.01 /
  Adam betas (0.9,0.98) eps 1e-9 wd=0.01).
torch.optim and clip_grad_norm_ are the same third-party arithmetic the
reference calls, so they are called here rather than restated.
"""
import time

import numpy as np
import torch

from . import encoder as enc
from . import losses as L


def get_losses(loss_name, pred, tgt_ang, tgt_crds, src_seq, pool=None, do_backwards=True,
               return_rmsd=False, eval_mode=False, combined_drmsd_weight=0.5):
    """train.py:57-111.  Returns the reference's dict of 10 entries."""
    m_full = L.mse_over_angles(pred, tgt_ang)
    m_bb = L.mse_over_angles(pred, tgt_ang, bb_only=True)
    m_sc = L.mse_over_angles(pred, tgt_ang, sc_only=True)
    rmsd = None
    if loss_name in ("lndrmsd", "drmsd", "combined") or eval_mode:
        ls = L.compute_batch_drmsd(pred, tgt_crds, src_seq, do_backward=do_backwards,
                                   retain_graph=loss_name == "combined", pool=pool,
                                   return_rmsd=return_rmsd)
        if return_rmsd:
            d, ln_d, d_bb, ln_bb, rmsd = ls
        else:
            d, ln_d, d_bb, ln_bb = ls
        c = L.combine_drmsd_mse(ln_d, m_full, w=combined_drmsd_weight)
        if loss_name == "lndrmsd":
            loss = ln_d
        elif loss_name == "drmsd":
            loss = d
        elif loss_name == "combined":
            loss = c
            if do_backwards:
                c.backward()
        else:
            loss = m_full
    else:
        d = ln_d = d_bb = ln_bb = c = torch.tensor(0)
        loss = m_full
        if do_backwards:
            m_full.backward()
    return {"loss": loss, "drmsd-full": d, "lndrmsd-full": ln_d, "drmsd-bb": d_bb,
            "lndrmsd-bb": ln_bb, "combined-full": c, "mse-full": m_full, "mse-bb": m_bb,
            "mse-sc": m_sc, "rmsd-full": rmsd}


def make_optimizer(params, name="sgd", lr=1e-4, weight_decay=True):
    # train.py:371-381
    wd = 10e-3 if weight_decay else 0
    if name == "adam":
        return torch.optim.Adam(params, betas=(0.9, 0.98), eps=1e-09, lr=lr, weight_decay=wd)
    return torch.optim.SGD(params, lr=lr, weight_decay=wd)


class CpuTrainer:
    """Holds leaf parameters + optimizer; `.step(batch)` = one train_epoch iteration."""

    def __init__(self, params, nhead, loss="drmsd", optimizer="sgd", lr=1e-4, clip=1.0,
                 weight_decay=True, pool=None):
        self.buffers = {k: v for k, v in params.items() if k.endswith(".pe")}
        self.params = {k: v.clone().requires_grad_() for k, v in params.items() if not k.endswith(".pe")}
        self.nhead, self.loss, self.clip, self.pool = nhead, loss, clip, pool
        self.opt = make_optimizer(list(self.params.values()), optimizer, lr, weight_decay)

    def all_tensors(self):
        return {**self.params, **self.buffers}

    def forward(self, src_seq):
        return enc.encoder_forward(self.all_tensors(), src_seq, self.nhead)

    def step(self, src_seq, tgt_ang, tgt_crds, keep_grads=False):
        """`keep_grads`: leave the step's losses, its gradients (as back-propagated, before the clip) and their global norm
        behind as `last_losses` / `last_grads` / `last_grad_norm` - what bench.py's parity block and the full-size parity test
        compare the HIP step with."""
        self.opt.zero_grad()
        pred = self.forward(src_seq)
        losses = get_losses(self.loss, pred, tgt_ang, tgt_crds, src_seq, pool=self.pool)
        if keep_grads:
            self.last_losses = losses
            self.last_grads = {k: p.grad.detach().clone() for k, p in self.params.items() if p.grad is not None}
            self.last_grad_norm = float(torch.sqrt(sum((g.double() ** 2).sum() for g in self.last_grads.values())))
        if self.clip:
            torch.nn.utils.clip_grad_norm_(list(self.params.values()), self.clip)
        self.opt.step()
        return losses


def time_cpu_steps(trainer, batch, n_steps=1, keep_grads=False):
    """Wall-clock residues/s of the CPU path, the metric of log.py:422-430."""
    src_seq = batch[0]
    n_res = int((src_seq != enc.PAD_ID).sum())
    t0 = time.perf_counter()
    for _ in range(n_steps):
        trainer.step(*batch, keep_grads=keep_grads)
    dt = (time.perf_counter() - t0) / n_steps
    return n_res / dt, dt
