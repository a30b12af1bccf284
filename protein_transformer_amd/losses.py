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
from .protein.structure_utils import get_backbone_from_full_coords  # noqa: F401  (losses.py:12: importable from here too)
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
def drmsd_forward_backward(pred_crd, true_crd, seq, need_grad=True, partial_budget_bytes=0):
    """Batched loss kernel. Returns stats [B,8] (device) and d(drmsd/n)/d(pred_crd) or None.  `partial_budget_bytes`: cap on the
    fixed-order partial sums of the pair sweep (0 = the library's 200 MB; smaller = the sweep runs in passes, same bits)."""
    _lib.require_gpu(pred_crd, true_crd, seq)
    B, L = seq.shape
    pred_crd, true_crd, seq = pred_crd.contiguous(), true_crd.contiguous(), seq.contiguous()
    stats = torch.empty(B, 8, dtype=torch.float32, device=seq.device)
    dcrd = torch.empty_like(pred_crd) if need_grad else None
    budget = int(partial_budget_bytes)
    nbytes = _lib.lib().ptamd_drmsd_workspace_bytes_budget(B, L, budget)
    ws = _lib.workspace("drmsd", nbytes, seq.device)
    rc = _lib.lib().ptamd_drmsd_fwd_bwd_budget(_lib.ptr(pred_crd), _lib.ptr(true_crd), _lib.ptr(seq), B, L, _lib.ptr(stats),
                                               _lib.ptr(dcrd), _lib.ptr(ws), ws.numel(), budget, _lib.stream())
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


def remove_sos_eos_from_input(input_seq):
    """A sequence of integers without a leading SOS / trailing EOS id (losses.py:39-46).  With the default vocabulary
    (no SOS / EOS characters) both ids are the unknown id 21 (protein/Sequence.py), as in the reference."""
    start_idx = 1 if input_seq[0] == VOCAB.sos_id else 0
    end_idx = -1 if input_seq[-1] == VOCAB.eos_id else None
    return input_seq[start_idx:end_idx]


def angles_to_coords(angles, seq, remove_batch_padding=False):
    """Torsional angles -> coordinates (losses.py:99-116)."""
    if remove_batch_padding:
        seq = seq[seq.ne(VOCAB.pad_id)]
    seq = remove_sos_eos_from_input(seq)
    angles = angles[:seq.shape[0]]
    return generate_coords(angles, seq)


def batch_loss(pred_sincos, true_crds, input_seqs, do_backward=True, return_crd=False):
    """Device-resident core of compute_batch_drmsd: no host synchronisation.

    Returns (stats [B,8] device tensor, d(sum_i lndrmsd_i)/d(pred_sincos) or None, status int32[1]) and, with
    `return_crd`, the predicted coordinates [B, L*14, 3] as a fourth value.
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
    return (stats, grad, status, crd) if return_crd else (stats, grad, status)


# ----------------------------------------------------------------------------- statistics hand-over
_PINNED = {}


def _pinned(kind, n, dtype, device):
    """Pinned host buffer per (kind, size, device, STREAM): a stream's reports are consumed in order (`LossReport.wait`
    runs before the next one is built), two streams never share one."""
    key = (kind, n, dtype, device, torch.cuda.current_stream(device).cuda_stream)
    buf = _PINNED.get(key)
    if buf is None:
        buf = _PINNED[key] = torch.empty(n, dtype=dtype).pin_memory()
    return buf


class LossReport:
    """The loss statistics of one batch on their way to the host, without draining the stream.

    The reference returns host numbers every step (losses.py:169-172, train.py:64-66).  Here the kernels leave
    per-protein dRMSD statistics [B,8], the six MSE sums, the status word and (evaluation) per-protein RMSDs on the
    device; this object enqueues ONE set of asynchronous copies into a pinned buffer right behind them, records an event,
    and `wait()` - called after the backward pass has been enqueued - blocks on that event only.

    Data parallel: every reported loss is a statistic of the GLOBAL batch, so the ranks first
    SUM-reduce a small fp64 vector (sums over proteins, protein / residue counts, MSE numerators and denominators, status
    flags): every rank then sees the same numbers and takes the same NaN / early-stopping / scheduler decisions.
    `global_mse_sums` is the device tensor the MSE gradient must be normalised with (count of the whole global batch).
    A rank whose shard is empty passes None for everything and still takes part in the reduction.
    """

    # [0:4] sums of drmsd, ln, bb, bb-ln  [4] proteins  [5] sum rmsd  [6:12] mse sums  [12:16] status bits  [16] residues
    # [17] proteins with rmsd  [18] ranks that passed a residue count (0: nobody counted - n_res stays None like on one rank)
    _NVEC = 19

    def __init__(self, device, stats=None, status=None, mse_sums_local=None, rmsd=None, n_res=None):
        from . import dp
        self.world = dp.world_size()
        self.n_res = n_res
        if self.world == 1:
            B = 0 if stats is None else stats.shape[0]
            self._B = B
            n = B * 8 + 6 + 1 + B
            buf = _pinned("report32", n, torch.float32, device)
            if stats is not None:
                buf[:B * 8].copy_(stats.reshape(-1), non_blocking=True)
            if mse_sums_local is not None:
                buf[B * 8:B * 8 + 6].copy_(mse_sums_local, non_blocking=True)
            if status is not None:       # raw int32 bits into the float slot
                buf.view(torch.int32)[B * 8 + 6:B * 8 + 7].copy_(status, non_blocking=True)
            if rmsd is not None:
                buf[B * 8 + 7:].copy_(rmsd, non_blocking=True)
            self._has = (stats is not None, mse_sums_local is not None, status is not None, rmsd is not None)
            self._buf = buf
            self.global_mse_sums = mse_sums_local
        else:
            v = torch.zeros(self._NVEC, dtype=torch.float64, device=device)
            if stats is not None:
                v[0:4] = stats[:, :4].double().sum(0)
                v[4] = stats.shape[0]
            if rmsd is not None:
                v[5] = rmsd.double().sum()
                v[17] = rmsd.shape[0]
            if mse_sums_local is not None:
                v[6:12] = mse_sums_local.double()
            if status is not None:
                v[12:16] = ((status.to(torch.int64) >> torch.arange(4, device=device)) & 1).double()
            v[16] = float(n_res or 0)
            v[18] = 0.0 if n_res is None else 1.0
            dp.all_reduce_sum_(v)
            self.global_mse_sums = v[6:12].float()
            buf = _pinned("report64", self._NVEC, torch.float64, device)
            buf.copy_(v, non_blocking=True)
            self._buf = buf
        self._event = torch.cuda.Event()
        self._event.record()

    def wait(self):
        """Block until the copies have landed; returns a dict of host numbers (float64 / int)."""
        self._event.synchronize()
        out = {"drmsd": 0.0, "lndrmsd": 0.0, "drmsd-bb": 0.0, "lndrmsd-bb": 0.0, "rmsd": None, "n_proteins": 0,
               "status": 0, "n_res": self.n_res, "mse": None}
        if self.world == 1:
            B = self._B
            host = self._buf.numpy()
            has_stats, has_mse, has_status, has_rmsd = self._has
            if has_stats and B:
                st = host[:B * 8].reshape(B, 8).astype(np.float64)
                out.update({"drmsd": np.mean(st[:, 0]), "lndrmsd": np.mean(st[:, 1]), "drmsd-bb": np.mean(st[:, 2]),
                            "lndrmsd-bb": np.mean(st[:, 3]), "n_proteins": B})
            if has_mse:
                out["mse"] = host[B * 8:B * 8 + 6].astype(np.float64)
            if has_status:
                out["status"] = int(host[B * 8 + 6:B * 8 + 7].view(np.int32)[0])
            if has_rmsd and B:
                out["rmsd"] = float(np.mean(host[B * 8 + 7:].astype(np.float64)))
        else:
            v = self._buf.numpy().copy()
            n = max(v[4], 1.0)
            out.update({"drmsd": v[0] / n, "lndrmsd": v[1] / n, "drmsd-bb": v[2] / n, "lndrmsd-bb": v[3] / n,
                        "n_proteins": int(v[4]), "mse": v[6:12],
                        "status": sum((1 << k) for k in range(4) if v[12 + k] > 0),
                        "n_res": int(v[16]) if v[18] > 0 else None})
            if v[17] > 0:
                out["rmsd"] = v[5] / v[17]
        return out


def _stats_to_host(stats, status):
    """Back-compatible helper: (pinned buffer of B*8 + 1 floats, event recorded behind the copy)."""
    n = stats.numel()
    buf = _pinned("stats", n + 1, torch.float32, stats.device)
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
    Returns np.mean over proteins of (drmsd, lndrmsd, bb drmsd, bb lndrmsd[, rmsd]); under data parallelism the means
    are those of the global batch (every rank calls this with its shard).
    """
    if backbone_only:
        raise NotImplementedError("--backbone_loss is broken in the reference too (SURVEY.md A-4)")
    dev = pred_angs.device
    true_crds, input_seqs = true_crds.to(dev), input_seqs.to(dev)
    stats, grad, status, crd = batch_loss(pred_angs, true_crds, input_seqs, do_backward, return_crd=True)
    rmsd = None
    if return_rmsd:
        from .eval_metrics import kabsch_rmsd_batch
        rmsd = kabsch_rmsd_batch(crd, true_crds, input_seqs)
    # The copy to the host is enqueued right behind the loss kernels and the host waits for THAT copy only after the
    # whole backward pass has been enqueued: waiting on the stream instead would drain the queue at the end of every
    # step and leave the GPU idle while the next launches are being issued.
    report = LossReport(dev, stats=stats, status=status, rmsd=rmsd)
    if do_backward:
        pred_angs.backward(gradient=grad.view_as(pred_angs), retain_graph=retain_graph)
    host = report.wait()
    raise_for_status(host["status"], theta_is_error=False)
    out = (host["drmsd"], host["lndrmsd"], host["drmsd-bb"], host["lndrmsd-bb"])
    if return_rmsd:
        out = out + (host["rmsd"],)
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
        from .eval_metrics import kabsch_rmsd_batch
        out = out + (float(kabsch_rmsd_batch(crd, crd_t, seq)[0]),)
    return out


# ----------------------------------------------------------------------------- angle MSE
def mse_sums(pred, true):
    """One pass over [B,L,24]: device tensor [6] = (sum, count) for full / backbone / side-chain columns, i.e. the three
    variants get_losses asks for back to back (train.py:64-66).  No host synchronisation."""
    _lib.require_gpu(pred, true)
    T = pred.shape[0] * pred.shape[1]
    out = torch.empty(6, dtype=torch.float32, device=pred.device)
    ws = _lib.workspace("mse_angles", _lib.lib().ptamd_mse_angles_workspace_bytes(), pred.device)
    rc = _lib.lib().ptamd_mse_angles_fwd(_lib.ptr(pred.detach().float().contiguous()),
                                         _lib.ptr(true.float().contiguous()), T, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                         _lib.stream())
    _lib.check(rc, "mse_angles_fwd")
    return out


def mse_grad(pred, true, sums, coef=1.0, accumulate_into=None):
    """d(coef * mse_full)/d(pred) = coef * 2 (pred - true) / count on the selected elements, where count = sums[1] is
    read ON THE DEVICE (under data parallelism it is the count of the global batch, so the SUM of the ranks' parameter
    gradients is the gradient of the global mean).  `accumulate_into`: add to that [B,L,24] tensor instead of a new one."""
    T = pred.shape[0] * pred.shape[1]
    acc = accumulate_into is not None
    dpred = accumulate_into if acc else torch.empty(pred.shape, dtype=torch.float32, device=pred.device)
    assert dpred.is_contiguous()
    rc = _lib.lib().ptamd_mse_angles_bwd(_lib.ptr(pred.detach().float().contiguous()), _lib.ptr(true.float().contiguous()),
                                         T, _lib.ptr(sums), float(coef), int(acc), _lib.ptr(dpred), _lib.stream())
    _lib.check(rc, "mse_angles_bwd")
    return dpred


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
        # API-parity path (a host read of the incoming scalar); the training step uses mse_grad directly
        return mse_grad(pred, true, sums, coef=float(g)).view_as(pred), None, None


def mse_over_angles(pred, true, bb_only=False, sc_only=False):
    """Mean squared error over (cos, sin) values with batch padding (all-zero rows) and missing
    angles (NaN) removed (losses.py:175-214).  Returns a 0-d device tensor."""
    assert len(pred.shape) == 3, "This function must operate on a batch of angles."
    if pred.shape[-1] != NUM_PREDICTED_ANGLES * 2:
        raise Exception("Unknown angle tensor shape.")
    which = 1 if bb_only else (2 if sc_only else 0)
    return _MseFn.apply(pred, true.to(pred.device), which)
