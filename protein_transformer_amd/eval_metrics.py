"""Eval-only RMSD after optimal superposition (Kabsch), SURVEY.md section 8f row 1.

Reference: `rmsd` (/root/reference/protein_transformer/losses.py:281-286) calls ProDy's
calcTransformation / calcRMSD, reached only from `eval_epoch` (train.py:125-127) with
return_rmsd=True.  ProDy is not installable here, so this is the textbook Kabsch algorithm
(PARITY UNPINNED, same caveat as oracle/losses.py:kabsch_rmsd).  The 3x3 SVDs are tiny; they run
through torch.linalg on the device as plumbing of the evaluation report, not of the training step.
"""
import numpy as np
import torch

from .protein.Structure import NUM_PREDICTED_COORDS


def rmsd_of_slots(pred_crd, true_crd):
    """pred_crd, true_crd [L*14,3] device tensors; atoms with NaN truth are skipped."""
    ok = ~torch.isnan(true_crd).any(dim=1)
    a, b = pred_crd[ok].double(), true_crd[ok].double()
    if a.shape[0] == 0:
        return float("nan")
    ac, bc = a - a.mean(0), b - b.mean(0)
    u, s, vt = torch.linalg.svd(ac.T @ bc)
    d = torch.sign(torch.linalg.det(u @ vt))
    e0 = (ac ** 2).sum() + (bc ** 2).sum()
    return float(torch.sqrt(torch.clamp(e0 - 2.0 * (s[0] + s[1] + d * s[2]), min=0.0) / a.shape[0]))


def batch_rmsd(pred_sincos, true_crds, input_seqs):
    """np.mean over proteins of the superposed RMSD of the structures built from pred_sincos."""
    from .losses import angles_forward
    from .protein.Sequence import VOCAB
    from .protein.Structure import nerf_forward
    B, L = input_seqs.shape
    ang = angles_forward(pred_sincos.detach().float().contiguous().view(B, L, -1))
    crd, _ = nerf_forward(ang, input_seqs)
    vals = []
    for b in range(B):
        n = int((input_seqs[b] != VOCAB.pad_id).sum()) * NUM_PREDICTED_COORDS
        vals.append(rmsd_of_slots(crd[b, :n], true_crds[b, :n]))
    return np.mean(vals)
