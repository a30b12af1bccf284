"""Oracle: atan2 transform, dRMSD, per-protein loss worker, batch driver, angle MSE.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates /root/reference/protein_transformer/losses.py:
  combine_drmsd_mse 15-23, inverse_trig_transform 26-36, drmsd_work 49-98,
  angles_to_coords 101-116, compute_batch_drmsd 133-172, mse_over_angles 175-214,
  pairwise_internal_dist 233-253, drmsd 256-278, rmsd 281-286 (Kabsch; unpinned),
and get_backbone_from_full_coords (protein/structure_utils.py:19-32).
"""
import numpy as np
import torch

from .geometry import NUM_ANGLES, NUM_SLOTS, PAD_ID, SC_ANGLE0, generate_coords


def combine_drmsd_mse(d, mse, w=.5, lndrmsd_norm=0.02, mse_norm=0.01):
    # losses.py:15-23 (the wandb.log side effect is not part of the arithmetic)
    d = w * (d / lndrmsd_norm)
    mse = (1 - w) * (mse / mse_norm)
    return d + mse


def inverse_trig_transform(t):
    # losses.py:26-36: [B, L, 24] (cos, sin interleaved) -> [B, L, 12] radians
    t = t.view(t.shape[0], -1, NUM_ANGLES, 2)
    return torch.atan2(t[:, :, :, 1], t[:, :, :, 0])


def pairwise_internal_dist(x):
    # losses.py:233-253: ||xi||^2 + ||xj||^2 - 2 xi.xj, clamp 1e-30, sqrt
    assert len(x.shape) == 2
    sq = x.pow(2).sum(dim=-1, keepdim=True)
    res = torch.addmm(sq.transpose(-2, -1), x, x.transpose(-2, -1), alpha=-2).add_(sq)
    return res.clamp_min_(1e-30).sqrt_()


def drmsd(a, b):
    # losses.py:256-278: sqrt(mean over i<j of (d_ij(a) - d_ij(b))^2), fp32
    a_ = pairwise_internal_dist(a)
    b_ = pairwise_internal_dist(b)
    i = torch.triu_indices(a_.shape[0], a_.shape[1], offset=1)
    mse = torch.nn.functional.mse_loss(a_[i[0], i[1]].float(), b_[i[0], i[1]].float())
    return torch.sqrt(mse)


def backbone_of(crds):
    # structure_utils.py:19-32: keep slots 0..2 (N, CA, C) of every 14
    mask = torch.tensor(([True] * 3 + [False] * (NUM_SLOTS - 3)) * (crds.shape[0] // NUM_SLOTS))
    return crds[mask, :]


def kabsch_rmsd(a, b):
    """RMSD of `a` optimally superposed on `b`.

    PARITY UNPINNED: the reference calls ProDy (losses.py:281-286), which is not
    installed in any environment available to this build; this is the textbook
    Kabsch superposition that ProDy's calcTransformation implements.
    """
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    ac, bc = a - a.mean(0), b - b.mean(0)
    u, s, vt = np.linalg.svd(ac.T @ bc)
    d = np.sign(np.linalg.det(u @ vt))
    e0 = (ac ** 2).sum() + (bc ** 2).sum()
    return float(np.sqrt(max(e0 - 2.0 * (s[0] + s[1] + d * s[2]), 0.0) / a.shape[0]))


def angles_to_coords(angles, seq, remove_batch_padding=False):
    # losses.py:101-116 (default vocabulary has no SOS/EOS, so no trimming)
    if remove_batch_padding:
        seq = seq[seq.ne(PAD_ID)]
    angles = angles[:seq.shape[0]]
    return generate_coords(angles, seq)


def drmsd_work(pred_ang, true_crd, input_seq, return_rmsd=False, do_backward=True):
    """One protein (losses.py:49-98).

    pred_ang [L_pad,12] radians, true_crd [L_pad*14,3] (NaN = missing),
    input_seq [L_pad] (trailing pad id 20).  Returns
    (grad [L_pad,12] or None, drmsd, drmsd/n, bb_drmsd, bb_drmsd/n_bb[, rmsd]).
    The backward is always of the length-normalised loss (losses.py:80,91-92).
    """
    pred_ang = torch.as_tensor(np.asarray(pred_ang)).clone()
    true_crd = torch.as_tensor(np.asarray(true_crd))
    input_seq = torch.as_tensor(np.asarray(input_seq))
    pred_ang.requires_grad_()
    leaf = pred_ang

    keep = input_seq.ne(PAD_ID)
    input_seq = input_seq[keep]
    true_crd = true_crd[:input_seq.shape[0] * NUM_SLOTS]

    pred_crd = angles_to_coords(pred_ang, input_seq)

    present = torch.isnan(true_crd).eq(0)
    pred_m = pred_crd[present].reshape(-1, 3)
    true_m = true_crd[present].reshape(-1, 3)
    loss = drmsd(pred_m, true_m)
    l_normed = loss / pred_m.shape[0]

    pred_bb, true_bb = backbone_of(pred_crd), backbone_of(true_crd)
    present_bb = torch.isnan(true_bb).eq(0)
    pred_bb_m = pred_bb[present_bb].reshape(-1, 3)
    true_bb_m = true_bb[present_bb].reshape(-1, 3)
    bb_loss = drmsd(pred_bb_m, true_bb_m)
    bb_normed = bb_loss / pred_bb_m.shape[0]

    if do_backward:
        l_normed.backward()
    out = (leaf.grad, loss.item(), l_normed.item(), bb_loss.item(), bb_normed.item())
    if return_rmsd:
        out = out + (kabsch_rmsd(pred_m.data.numpy(), true_m.data.numpy()),)
    return out


def compute_batch_drmsd(pred_angs, true_crds, input_seqs, return_rmsd=False,
                        do_backward=False, retain_graph=False, pool=None):
    """Batch driver (losses.py:133-172).

    pred_angs [B,L,24] (cos,sin pairs; may require grad), true_crds [B,L*14,3],
    input_seqs [B,L].  Injects sum_i d(lndrmsd_i)/d(angles) into the graph of
    pred_angs (a SUM over proteins, losses.py:166-167) and returns the np.mean
    of (drmsd, lndrmsd, bb drmsd, bb lndrmsd[, rmsd]) over proteins.
    `pool`: an optional multiprocessing pool with .map (losses.py:144-147).
    """
    pred = inverse_trig_transform(pred_angs.cpu())
    true_crds, input_seqs = true_crds.cpu(), input_seqs.cpu()
    jobs = [(a.detach().numpy(), c.detach().numpy(), s.detach().numpy(), return_rmsd, do_backward)
            for a, c, s in zip(pred, true_crds, input_seqs)]
    if pool is not None:
        results = pool.map(_drmsd_work_star, jobs)
    else:
        results = [_drmsd_work_star(j) for j in jobs]
    grads = [r[0] for r in results]
    cols = list(zip(*[r[1:] for r in results]))
    if do_backward:
        pred.backward(gradient=torch.stack(grads), retain_graph=retain_graph)
    return tuple(np.mean(c) for c in cols)


def _drmsd_work_star(job):
    return drmsd_work(*job)


def mse_over_angles(pred, true, bb_only=False, sc_only=False):
    """losses.py:175-214 on [B,L,24] (cos,sin) tensors.

    Rows where the truth is all zero are batch padding; NaN truth elements are
    missing angles.  bb = first 12 of 24 columns, sc = last 12.
    """
    assert len(pred.shape) == 3
    width = pred.shape[-1]
    per = 2 if width == NUM_ANGLES * 2 else 1
    if bb_only:
        pred, true = pred[:, :, :SC_ANGLE0 * per], true[:, :, :SC_ANGLE0 * per]
    elif sc_only:
        pred, true = pred[:, :, SC_ANGLE0 * per:], true[:, :, SC_ANGLE0 * per:]
    rows = true.ne(0).any(dim=2)
    t = true[rows]
    ok = torch.isnan(t).eq(0)
    return torch.nn.functional.mse_loss(pred[rows][ok], t[ok])
