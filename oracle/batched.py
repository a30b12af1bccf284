"""Oracle (second, faster checker): the same NeRF build vectorised over the batch.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Same formulas as oracle/geometry.py (reference: protein/Structure.py:23-65,
protein/StructureBuilder.py:147-231) but every placement is one torch op over
all B proteins, so full-size batches (B=32, L=512) finish in seconds, in fp32
or fp64, with autograd providing the gradients.  It is itself checked against
oracle/geometry.py (tests/test_oracle_golden.py), i.e. transitively against
the golden vectors captured from the reference.
"""
import math

import torch

from .geometry import (BA_CA_C_O, BL_CA_C, BL_C_N, BL_C_O, BL_N_CA, NUM_SLOTS, PAD_ID,
                       SC_ANGLE0, SC_PROGRAM)

_MAX_SC = 10


def _tables(dtype):
    bond = torch.zeros(21, _MAX_SC, dtype=dtype)
    angle = torch.zeros(21, _MAX_SC, dtype=dtype)
    kind = torch.zeros(21, _MAX_SC, dtype=torch.long)      # 0 absent, 1 'p', 2 'i', 3 const
    const = torch.zeros(21, _MAX_SC, dtype=dtype)
    par = torch.zeros(21, _MAX_SC, 3, dtype=torch.long)
    for r, prog in SC_PROGRAM.items():
        for k, (b, a, t, p) in enumerate(prog):
            # constants are created as fp32 tensors by the reference (StructureBuilder.py:250-252)
            bond[r, k] = float(torch.tensor(b, dtype=torch.float32)) if dtype == torch.float32 else b
            angle[r, k] = float(torch.tensor(a, dtype=torch.float32)) if dtype == torch.float32 else a
            if t == "p":
                kind[r, k] = 1
            elif t == "i":
                kind[r, k] = 2
            else:
                kind[r, k] = 3
                const[r, k] = t
            if p is not None:
                par[r, k] = torch.tensor(p)
    return bond, angle, kind, const, par


def _unit(v):
    return v / v.norm(dim=-1, keepdim=True).clamp_min(1e-12)


def nerf_b(a, b, c, l, theta, chi):
    """Batched placement: a,b,c [B,3]; l, theta, chi [B] (or scalars)."""
    w = _unit(b - a)
    x = _unit(c - b)
    z = _unit(torch.linalg.cross(w, x))
    y = torch.linalg.cross(z, x)
    d0 = -l * torch.cos(theta)
    d1 = l * torch.sin(theta) * torch.cos(chi)
    d2 = l * torch.sin(theta) * torch.sin(chi)
    return c + d0[:, None] * x + d1[:, None] * y + d2[:, None] * z


def generate_coords_batched(ang, seq, dtype=torch.float32):
    """ang [B,L,12] radians, seq [B,L] ids (pad 20 trailing) -> [B, L*14, 3].

    Padded residues produce zeros.  Lengths may differ per protein.
    """
    B, L, _ = ang.shape
    ang = ang.to(dtype)
    bond, angle, kind, const, par = _tables(dtype)
    lens = (seq != PAD_ID).sum(1)
    pi = math.pi
    full = lambda v: torch.full((B,), v, dtype=dtype)
    zeros3 = torch.zeros(B, 3, dtype=dtype)

    bb = []
    n0 = torch.tensor([0, 0, 0.001], dtype=dtype).expand(B, 3)
    ca0 = n0 + torch.tensor([BL_N_CA, 0, 0], dtype=dtype)
    a03 = ang[:, 0, 3].detach()
    c0 = ca0 + torch.stack([torch.cos(pi - a03) * BL_CA_C, torch.sin(pi - a03) * BL_CA_C,
                            torch.zeros(B, dtype=dtype)], dim=1)
    o0 = nerf_b(n0, ca0, c0, full(BL_C_O), full(BA_CA_C_O), ang[:, 0, 1] - pi)
    bb.append((n0, ca0, c0, o0))
    for i in range(1, L):
        pn, pca, pc, _ = bb[-1]
        pa, a = ang[:, i - 1], ang[:, i]
        n = nerf_b(pn, pca, pc, full(BL_C_N), pa[:, 4], pa[:, 1])
        ca = nerf_b(pca, pc, n, full(BL_N_CA), pa[:, 5], pa[:, 2])
        c = nerf_b(pc, n, ca, full(BL_CA_C), a[:, 3], a[:, 0])
        o = nerf_b(n, ca, c, full(BL_C_O), full(BA_CA_C_O), a[:, 1] - pi)
        bb.append((n, ca, c, o))

    rows = []
    for i in range(L):
        live = (i < lens)
        res = seq[:, i].clamp(max=20)
        res = torch.where(live, res, torch.full_like(res, 20))
        slots = list(bb[i]) + [zeros3] * (NUM_SLOTS - 4)
        last = torch.zeros(B, dtype=dtype)
        for k in range(_MAX_SC):
            kd = kind[res, k]
            if not bool((kd > 0).any()):
                break
            if k == 0:
                if i == 0:
                    pa_, pb_, pc_ = bb[1][0] if L > 1 else zeros3, slots[2], slots[1]
                else:
                    pa_, pb_, pc_ = bb[i - 1][2], slots[0], slots[1]
            else:
                st = torch.stack(slots, dim=1)                      # [B,14,3]
                idx = par[res, k]                                   # [B,3]
                g = st.gather(1, idx[:, :, None].expand(B, 3, 3))
                pa_, pb_, pc_ = g[:, 0], g[:, 1], g[:, 2]
            chi = torch.where(kd == 1, ang[:, i, SC_ANGLE0 + min(k, 5)],
                              torch.where(kd == 2, last - pi, const[res, k]))
            present = (kd > 0)
            # keep absent lanes finite so autograd never sees NaN
            pa_s = torch.where(present[:, None], pa_, torch.tensor([1., 0, 0], dtype=dtype).expand(B, 3))
            pb_s = torch.where(present[:, None], pb_, torch.tensor([0., 1, 0], dtype=dtype).expand(B, 3))
            pc_s = torch.where(present[:, None], pc_, torch.tensor([0., 0, 1], dtype=dtype).expand(B, 3))
            pt = nerf_b(pa_s, pb_s, pc_s, bond[res, k], angle[res, k], chi)
            slots[4 + k] = torch.where(present[:, None], pt, zeros3)
            last = chi
        st = torch.stack(slots, dim=1)
        rows.append(torch.where(live[:, None, None], st, torch.zeros_like(st)))
    return torch.stack(rows, dim=1).reshape(B, L * NUM_SLOTS, 3)


def drmsd_direct(a, b):
    """sqrt(mean_{i<j} (|ai-aj| - |bi-bj|)^2) by direct differences, in a's dtype."""
    da = torch.cdist(a, a, compute_mode="donot_use_mm_for_euclid_dist")
    db = torch.cdist(b, b, compute_mode="donot_use_mm_for_euclid_dist")
    iu = torch.triu_indices(a.shape[0], a.shape[0], offset=1)
    return torch.sqrt(((da[iu[0], iu[1]] - db[iu[0], iu[1]]) ** 2).mean())


def batch_loss_and_grads(ang, seq, true_crd, dtype=torch.float64):
    """Fast full-batch checker: per-protein (drmsd, ln, bb, bb_ln), coords, d(sum ln)/d(ang).

    ang [B,L,12] radians; true_crd [B,L*14,3] with NaN for missing atoms.
    """
    ang = ang.detach().to(dtype).clone().requires_grad_()
    crd = generate_coords_batched(ang, seq, dtype)
    stats, total = [], 0.
    B, L = seq.shape
    for b in range(B):
        n_res = int((seq[b] != PAD_ID).sum())
        t = true_crd[b, :n_res * NUM_SLOTS].to(dtype)
        p = crd[b, :n_res * NUM_SLOTS]
        ok = ~torch.isnan(t).any(dim=1)
        d = drmsd_direct(p[ok], t[ok])
        ln = d / int(ok.sum())
        slot = torch.arange(n_res * NUM_SLOTS) % NUM_SLOTS
        okb = ok & (slot < 3)
        dbb = drmsd_direct(p[okb], t[okb])
        stats.append((d.item(), ln.item(), dbb.item(), (dbb / int(okb.sum())).item(), int(ok.sum()), int(okb.sum())))
        total = total + ln
    total.backward()
    return stats, crd.detach(), ang.grad.detach()
