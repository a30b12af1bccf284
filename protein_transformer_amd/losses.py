"""Loss functions of the training hot path, on the device.

Same names, arguments and return conventions as
/root/reference/protein_transformer/losses.py (combine_drmsd_mse :15, inverse_trig_transform :26,
drmsd_work :49, angles_to_coords :101, compute_batch_drmsd :133, mse_over_angles :175, drmsd :256);
the arithmetic runs in csrc/geometry.hip and csrc/drmsd.hip through libptamd.  Nothing is moved
to the CPU and no worker pool is needed: `device` and `pool` are accepted and ignored.
"""
import numpy as np
import torch

from . import _lib
from .protein.Sequence import VOCAB
from .protein.Structure import (NUM_PREDICTED_ANGLES, NUM_PREDICTED_COORDS, SC_ANGLES_START_POS, generate_coords,
                                nerf_backward, nerf_forward, raise_for_status)


def combine_drmsd_mse(d, mse, w=.5, lndrmsd_norm=0.02, mse_norm=0.01, log=True):
    """w * d / lndrmsd_norm + (1 - w) * mse / mse_norm   (losses.py:15-23; `log` only fed wandb)."""
    d = w * (d / lndrmsd_norm)
    mse = (1 - w) * (mse / mse_norm)
    return d + mse


# ----------------------------------------------------------------------------- atan2
def angles_forward(sincos):
    _lib.require_gpu(sincos)
    sincos = sincos.contiguous()
    ang = torch.empty(sincos.shape[:-1] + (sincos.shape[-1] // 2,), dtype=torch.float32, device=sincos.device)
    rc = _lib.lib().ptamd_angles_fwd(_lib.ptr(sincos), _lib.ptr(ang), ang.numel(), _lib.stream())
    _lib.check(rc, "angles_fwd")
    return ang


def angles_backward(sincos, dang):
    dsc = torch.empty_like(sincos)
    rc = _lib.lib().ptamd_angles_bwd(_lib.ptr(sincos.contiguous()), _lib.ptr(dang.contiguous()), _lib.ptr(dsc),
                                     dang.numel(), _lib.stream())
    _lib.check(rc, "angles_bwd")
    return dsc


class _AnglesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sincos):
        ctx.save_for_backward(sincos)
        return angles_forward(sincos)

    @staticmethod
    def backward(ctx, dang):
        (sincos,) = ctx.saved_tensors
        return angles_backward(sincos, dang)


def inverse_trig_transform(t):
    """[B, L, 24] (cos, sin interleaved) -> [B, L, 12] radians via atan2 (losses.py:26-36)."""
    t = t.view(t.shape[0], -1, NUM_PREDICTED_ANGLES * 2)
    return _AnglesFn.apply(t.float())


# ----------------------------------------------------------------------------- dRMSD
def drmsd_forward_backward(pred_crd, true_crd, seq, need_grad=True):
    """Batched loss kernel. Returns stats [B,8] (device) and d(drmsd/n)/d(pred_crd) or None."""
    _lib.require_gpu(pred_crd, true_crd, seq)
    B, L = seq.shape
    pred_crd, true_crd, seq = pred_crd.contiguous(), true_crd.contiguous(), seq.contiguous()
    stats = torch.empty(B, 8, dtype=torch.float32, device=seq.device)
    dcrd = torch.empty_like(pred_crd) if need_grad else None
    nbytes = _lib.lib().ptamd_drmsd_workspace_bytes(B, L)
    ws = _lib.workspace("drmsd", nbytes, seq.device)
    rc = _lib.lib().ptamd_drmsd_fwd_bwd(_lib.ptr(pred_crd), _lib.ptr(true_crd), _lib.ptr(seq), B, L, _lib.ptr(stats),
                                        _lib.ptr(dcrd), _lib.ptr(ws), ws.numel(), _lib.stream())
    _lib.check(rc, "drmsd_fwd_bwd")
    return stats, dcrd


class _DrmsdFn(torch.autograd.Function):
    """drmsd(a, b) for two [n,3] point sets, differentiable in a."""

    @staticmethod
    def forward(ctx, a, b):
        n = a.shape[0]
        L = max(2, -(-n // NUM_PREDICTED_COORDS))
        dev = a.device
        pa = torch.zeros(1, L * NUM_PREDICTED_COORDS, 3, dtype=torch.float32, device=dev)
        pb = torch.full((1, L * NUM_PREDICTED_COORDS, 3), float("nan"), dtype=torch.float32, device=dev)
        pa[0, :n], pb[0, :n] = a.float(), b.float()
        seq = torch.zeros(1, L, dtype=torch.int64, device=dev)
        stats, dcrd = drmsd_forward_backward(pa, pb, seq, need_grad=True)
        ctx.save_for_backward(dcrd)
        ctx.n = n
        return stats[0, 0].clone()

    @staticmethod
    def backward(ctx, g):
        (dcrd,) = ctx.saved_tensors
        return g * ctx.n * dcrd[0, :ctx.n], None      # kernel differentiates drmsd / n


def drmsd(a, b):
    """Distance RMSD between coordinate tensors a and b, both [n,3] (losses.py:256-278)."""
    dev = a.device if a.is_cuda else torch.device("cuda", torch.cuda.current_device())
    return _DrmsdFn.apply(a.to(dev), b.to(dev))


def pairwise_internal_dist(x):
    """All pairwise distances of an [n, d] coordinate tensor -> [n, n] (losses.py:233-253).  API parity only:
    the loss kernels never build this matrix."""
    assert len(x.shape) == 2, "Pairwise internal distance method is not implemented for batches."
    dev = x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device())
    x = x.to(dev, torch.float32).contiguous()
    out = torch.empty(x.shape[0], x.shape[0], dtype=torch.float32, device=dev)
    rc = _lib.lib().ptamd_pairwise_dist(_lib.ptr(x), x.shape[0], x.shape[1], _lib.ptr(out), _lib.stream())
    _lib.check(rc, "pairwise_dist")
    return out


def angles_to_coords(angles, seq, remove_batch_padding=False):
    """Torsional angles -> coordinates (losses.py:101-116)."""
    if remove_batch_padding:
        seq = seq[seq.ne(VOCAB.pad_id)]
    angles = angles[:seq.shape[0]]
    return generate_coords(angles, seq)


def batch_loss(pred_sincos, true_crds, input_seqs, do_backward=True):
    """Device-resident core of compute_batch_drmsd: no host synchronisation.

    Returns (stats [B,8] device tensor, d(sum_i lndrmsd_i)/d(pred_sincos) or None, status int32[1]).
    """
    pred_sincos = pred_sincos.detach().float().contiguous()
    B, L = input_seqs.shape
    sc = pred_sincos.view(B, L, NUM_PREDICTED_ANGLES * 2)
    ang = angles_forward(sc)
    crd, status = nerf_forward(ang, input_seqs)
    stats, dcrd = drmsd_forward_backward(crd, true_crds.float(), input_seqs, need_grad=do_backward)
    grad = None
    if do_backward:
        dang = nerf_backward(ang, input_seqs, crd, dcrd)
        grad = angles_backward(sc, dang)
    return stats, grad, status


_HOST_STATS = {}


def _stats_to_host(stats, status):
    """Asynchronous copy of the per-protein statistics [B,8] and the status word into a (cached) pinned buffer;
    returns (buffer of B*8 + 1 floats, event recorded behind the copy)."""
    n = stats.numel()
    buf = _HOST_STATS.get(n)
    if buf is None:
        buf = _HOST_STATS[n] = torch.empty(n + 1, dtype=torch.float32).pin_memory()
    buf[:n].copy_(stats.reshape(-1), non_blocking=True)
    buf[n:].copy_(status.float(), non_blocking=True)   # a handful of flag bits: exact in fp32
    done = torch.cuda.Event()
    done.record()
    return buf, done


def compute_batch_drmsd(pred_angs, true_crds, input_seqs, device=None, return_rmsd=False,
                        do_backward=False, retain_graph=False, pool=None, backbone_only=False):
    """DRMSD loss of a batch (losses.py:133-172), entirely on the GPU.

    pred_angs [B,L,24] (cos,sin) predictions attached to the model's graph, true_crds [B,L*14,3]
    (NaN = missing atom), input_seqs [B,L].  With do_backward the SUM over proteins of
    d(lndrmsd_i)/d(pred_angs) is back-propagated through pred_angs (losses.py:166-167).
    Returns np.mean over proteins of (drmsd, lndrmsd, bb drmsd, bb lndrmsd[, rmsd]).
    """
    if backbone_only:
        raise NotImplementedError("--backbone_loss is broken in the reference too (SURVEY.md A-4)")
    dev = pred_angs.device
    stats, grad, status = batch_loss(pred_angs, true_crds.to(dev), input_seqs.to(dev), do_backward)
    # The reference returns host numbers every step.  The copy is enqueued right behind the loss kernels and the host
    # waits for THAT copy only after the whole backward pass has been enqueued: waiting on the stream instead would
    # drain the queue at the end of every step and leave the GPU idle while the next launches are being issued.
    host_buf, copied = _stats_to_host(stats, status)
    if do_backward:
        pred_angs.backward(gradient=grad.view_as(pred_angs), retain_graph=retain_graph)
    copied.synchronize()
    host = host_buf[:-1].view(-1, 8).numpy().astype(np.float64)
    raise_for_status(int(host_buf[-1].item()), theta_is_error=False)
    out = (np.mean(host[:, 0]), np.mean(host[:, 1]), np.mean(host[:, 2]), np.mean(host[:, 3]))
    if return_rmsd:
        from .eval_metrics import batch_rmsd
        out = out + (batch_rmsd(pred_angs, true_crds.to(dev), input_seqs.to(dev)),)
    return out


def drmsd_work(pred_ang, true_crd, input_seq, return_rmsd=False, do_backward=True, backbone_only=False):
    """One protein (losses.py:49-98): returns (grad [L,12] or None, drmsd, lndrmsd, bb, bb_ln[, rmsd])."""
    if backbone_only:
        raise NotImplementedError("--backbone_loss is broken in the reference too (SURVEY.md A-4)")
    dev = torch.device("cuda", torch.cuda.current_device())
    ang = torch.as_tensor(np.asarray(pred_ang) if not torch.is_tensor(pred_ang) else pred_ang).to(dev, torch.float32)[None]
    crd_t = torch.as_tensor(np.asarray(true_crd) if not torch.is_tensor(true_crd) else true_crd).to(dev, torch.float32)[None]
    seq = torch.as_tensor(np.asarray(input_seq) if not torch.is_tensor(input_seq) else input_seq).to(dev, torch.int64)[None]
    crd, status = nerf_forward(ang.contiguous(), seq)
    stats, dcrd = drmsd_forward_backward(crd, crd_t, seq, need_grad=do_backward)
    grad = nerf_backward(ang.contiguous(), seq, crd, dcrd)[0].cpu() if do_backward else None
    raise_for_status(int(status.item()), theta_is_error=False)
    s = stats[0].cpu().numpy().astype(np.float64)
    out = (grad, float(s[0]), float(s[1]), float(s[2]), float(s[3]))
    if return_rmsd:
        from .eval_metrics import rmsd_of_slots
        out = out + (rmsd_of_slots(crd[0], crd_t[0]),)
    return out


# ----------------------------------------------------------------------------- angle MSE
_mse_cache = (None, None, None, None)


def mse_sums(pred, true):
    """One pass over [B,L,24]: device tensor [6] = (sum, count) for full / backbone / side-chain columns.
    get_losses asks for the three variants back to back (train.py:64-66); the pass runs once per (pred, true)."""
    global _mse_cache
    _lib.require_gpu(pred, true)
    key = (pred.data_ptr(), pred._version, true.data_ptr(), true._version)
    if _mse_cache[0] == key and _mse_cache[1] is pred and _mse_cache[2] is true:
        return _mse_cache[3]
    T = pred.shape[0] * pred.shape[1]
    out = torch.empty(6, dtype=torch.float32, device=pred.device)
    ws = _lib.workspace("mse_angles", _lib.lib().ptamd_mse_angles_workspace_bytes(), pred.device)
    rc = _lib.lib().ptamd_mse_angles_fwd(_lib.ptr(pred.detach().float().contiguous()),
                                         _lib.ptr(true.float().contiguous()), T, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                         _lib.stream())
    _lib.check(rc, "mse_angles_fwd")
    _mse_cache = (key, pred, true, out)
    return out


class _MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, true, which):
        sums = mse_sums(pred, true)
        ctx.save_for_backward(pred, true, sums)
        ctx.which = which
        return sums[2 * which] / sums[2 * which + 1]

    @staticmethod
    def backward(ctx, g):
        pred, true, sums = ctx.saved_tensors
        if ctx.which != 0:
            raise NotImplementedError("only the full-angle MSE is ever differentiated (train.py:86,97)")
        T = pred.shape[0] * pred.shape[1]
        dpred = torch.empty_like(pred, dtype=torch.float32)
        rc = _lib.lib().ptamd_mse_angles_bwd(_lib.ptr(pred.detach().float().contiguous()),
                                             _lib.ptr(true.float().contiguous()), T, _lib.ptr(sums), float(g), 0,
                                             _lib.ptr(dpred), _lib.stream())
        _lib.check(rc, "mse_angles_bwd")
        return dpred.view_as(pred), None, None


def mse_over_angles(pred, true, bb_only=False, sc_only=False):
    """Mean squared error over (cos, sin) values with batch padding (all-zero rows) and missing
    angles (NaN) removed (losses.py:175-214).  Returns a 0-d device tensor."""
    assert len(pred.shape) == 3, "This function must operate on a batch of angles."
    if pred.shape[-1] != NUM_PREDICTED_ANGLES * 2:
        raise Exception("Unknown angle tensor shape.")
    which = 1 if bb_only else (2 if sc_only else 0)
    return _MseFn.apply(pred, true.to(pred.device), which)
