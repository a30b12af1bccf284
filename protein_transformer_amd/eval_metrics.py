"""Eval-only RMSD after optimal superposition (Kabsch), SURVEY.md section 8f row 1.

Reference: `rmsd` (/root/reference/protein_transformer/losses.py:281-286) calls ProDy's
calcTransformation / calcRMSD, reached only from `eval_epoch` (train.py:125-127) with
return_rmsd=True.  ProDy is not installable here, so this is the textbook Kabsch algorithm
(PARITY UNPINNED, same caveat as oracle/losses.py:kabsch_rmsd).  It runs as ONE kernel launch per batch
(csrc/kabsch.hip: fp64 moments per protein, Jacobi eigenvalues of H^T H on the device) - no per-protein host
round trip, no torch.linalg.
"""
import numpy as np
import torch

from . import _lib
from .protein.Structure import NUM_PREDICTED_COORDS


def kabsch_rmsd_batch(pred_crd, true_crd, seq):
    """pred_crd, true_crd [B, L*14, 3] device tensors (NaN truth = absent atom), seq [B, L] -> rmsd [B] (device, no sync)."""
    _lib.require_gpu(pred_crd, true_crd, seq)
    B, L = seq.shape
    assert pred_crd.shape == (B, L * NUM_PREDICTED_COORDS, 3) and true_crd.shape == pred_crd.shape
    out = torch.empty(B, dtype=torch.float32, device=seq.device)
    rc = _lib.lib().ptamd_kabsch_rmsd(_lib.ptr(pred_crd.float().contiguous()), _lib.ptr(true_crd.float().contiguous()),
                                      _lib.ptr(seq.contiguous()), B, L, _lib.ptr(out), _lib.stream())
    _lib.check(rc, "kabsch_rmsd")
    return out


def rmsd(a, b):
    """RMSD between two [n,3] coordinate sets after superposing `a` on `b` (losses.py:281-286); host float."""
    dev = a.device if torch.is_tensor(a) and a.is_cuda else torch.device("cuda", torch.cuda.current_device())
    a = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).to(dev, torch.float32)
    b = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).to(dev, torch.float32)
    n = a.shape[0]
    L = max(1, -(-n // NUM_PREDICTED_COORDS))
    pa = torch.zeros(1, L * NUM_PREDICTED_COORDS, 3, dtype=torch.float32, device=dev)
    pb = torch.full((1, L * NUM_PREDICTED_COORDS, 3), float("nan"), dtype=torch.float32, device=dev)
    pa[0, :n], pb[0, :n] = a, b
    return float(kabsch_rmsd_batch(pa, pb, torch.zeros(1, L, dtype=torch.int64, device=dev))[0])


def batch_rmsd(pred_sincos, true_crds, input_seqs):
    """np.mean over proteins of the superposed RMSD of the structures built from pred_sincos (one host read)."""
    from .losses import angles_forward
    from .protein.Structure import nerf_forward
    B, L = input_seqs.shape
    ang = angles_forward(pred_sincos.detach().float().contiguous().view(B, L, -1))
    crd, _ = nerf_forward(ang, input_seqs)
    return float(kabsch_rmsd_batch(crd, true_crds, input_seqs).double().mean())
