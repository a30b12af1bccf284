"""Host-side mirror of the reference's `protein/Structure.py` over libptamd.

Same names and argument meaning as /root/reference/protein_transformer/protein/Structure.py:
constants :4-9, `generate_coords` :12-20.  The NeRF arithmetic itself (nerf :23-65 and
StructureBuilder.py) runs in the HIP kernels of csrc/geometry.hip.
"""
import torch

from .. import _lib
from .Sequence import VOCAB

NUM_PREDICTED_ANGLES = 12
NUM_PREDICTED_COORDS = 14
NUM_BB_TORSION_ANGLES = 3
NUM_BB_OTHER_ANGLES = 3
NUM_SC_ANGLES = NUM_PREDICTED_ANGLES - (NUM_BB_OTHER_ANGLES + NUM_BB_TORSION_ANGLES)
SC_ANGLES_START_POS = NUM_BB_OTHER_ANGLES + NUM_BB_TORSION_ANGLES


def raise_for_status(word, theta_is_error=True):
    """Turn the device status word into the exception the reference would have raised."""
    if word & _lib.ST_BAD_RESIDUE:
        raise KeyError("residue id outside 0..19 in a sequence (Sequence.py:50-51)")
    if word & _lib.ST_TOO_SHORT:
        raise StopIteration("a structure needs at least two residues (StructureBuilder.py:58-59)")
    if theta_is_error and (word & _lib.ST_BAD_THETA):
        raise AssertionError("theta must be in radians and in [-pi, pi] (Structure.py:42)")


_STATUS_POOL = {}


def _status_word(device):
    """A zeroed int32 word for the status bits of one NeRF launch, cut from a block that is zeroed ONCE per 1024 launches (a
    `torch.zeros(1)` per training step was one fill kernel per step); blocks stay alive through the words handed out."""
    key = (device.type, device.index)
    blk, used = _STATUS_POOL.get(key, (None, 0))
    if blk is None or used >= blk.numel():
        blk, used = torch.zeros(1024, dtype=torch.int32, device=device), 0
    _STATUS_POOL[key] = (blk, used + 1)
    return blk[used:used + 1]


def nerf_forward(ang, seq, status=None):
    """ang [B,L,12] fp32 cuda (radians), seq [B,L] int64 cuda -> crd [B,L*14,3]; no sync."""
    _lib.require_gpu(ang, seq)
    B, L, _ = ang.shape
    ang = ang.contiguous()
    seq = seq.contiguous()
    crd = torch.empty(B, L * NUM_PREDICTED_COORDS, 3, dtype=torch.float32, device=ang.device)
    if status is None:
        status = _status_word(ang.device)
    rc = _lib.lib().ptamd_nerf_fwd(_lib.ptr(ang), _lib.ptr(seq), B, L, _lib.ptr(crd), _lib.ptr(status), _lib.stream())
    _lib.check(rc, "nerf_fwd")
    return crd, status


def nerf_backward(ang, seq, crd, dcrd):
    """Adjoint of nerf_forward: dcrd [B,L*14,3] -> dang [B,L,12]."""
    _lib.require_gpu(ang, seq, crd, dcrd)
    B, L, _ = ang.shape
    dang = torch.empty_like(ang)
    nbytes = _lib.lib().ptamd_nerf_workspace_bytes(B, L)
    ws = _lib.workspace("nerf", nbytes, ang.device)
    rc = _lib.lib().ptamd_nerf_bwd(_lib.ptr(ang.contiguous()), _lib.ptr(seq.contiguous()), _lib.ptr(crd.contiguous()),
                                   _lib.ptr(dcrd.contiguous()), B, L, _lib.ptr(dang), _lib.ptr(ws), ws.numel(),
                                   _lib.stream())
    _lib.check(rc, "nerf_bwd")
    return dang


class _NerfFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ang, seq):
        crd, status = nerf_forward(ang, seq)
        ctx.save_for_backward(ang, seq, crd)
        ctx.mark_non_differentiable(status)
        return crd, status

    @staticmethod
    def backward(ctx, dcrd, _):
        ang, seq, crd = ctx.saved_tensors
        return nerf_backward(ang, seq, crd, dcrd), None


def generate_coords(angles, input_seq, device=None):
    """A protein's [L*14, 3] coordinates from its [L,12] angles and [L] sequence (Structure.py:12-20).

    `input_seq` may be a 1-letter string or an integer tensor without padding.  Differentiable
    with respect to `angles`.  `device` is kept for signature compatibility: the build always
    runs on the GPU the angles live on (or cuda:0 for CPU inputs, returning on that GPU).
    """
    if isinstance(input_seq, str):
        input_seq = torch.tensor([VOCAB._char2int[s] for s in input_seq])
    dev = angles.device if angles.is_cuda else torch.device("cuda", torch.cuda.current_device())
    ang = angles.to(dev, torch.float32)
    seq = input_seq.to(dev, torch.int64)
    crd, status = _NerfFn.apply(ang[None], seq[None])
    raise_for_status(int(status.item()))
    return crd[0]


def nerf(a, b, c, l, theta, chi):
    """Natural extension reference frame: place atom d from a, b, c with bond length l, bond angle theta and
    torsion chi (Structure.py:23-65).  Inputs may be single points ([3] / scalars) or batches ([n,3] / [n])."""
    dev = torch.device("cuda", torch.cuda.current_device())
    f = lambda t: torch.as_tensor(t, dtype=torch.float32).to(dev)                # noqa: E731
    a, b, c, l, theta, chi = f(a), f(b), f(c), f(l), f(theta), f(chi)
    single = a.dim() == 1
    a, b, c = (t.reshape(-1, 3).contiguous() for t in (a, b, c))
    n = a.shape[0]
    l, theta, chi = (t.reshape(-1).expand(n).contiguous() for t in (l, theta, chi))
    d = torch.empty(n, 3, dtype=torch.float32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = _lib.lib().ptamd_nerf_place(_lib.ptr(a), _lib.ptr(b), _lib.ptr(c), _lib.ptr(l), _lib.ptr(theta), _lib.ptr(chi),
                                     n, _lib.ptr(d), _lib.ptr(status), _lib.stream())
    _lib.check(rc, "nerf_place")
    raise_for_status(int(status.item()))
    return d[0] if single else d
