"""Deterministic synthetic proteins for benchmarks and tests.

There is no dataset on the box (the reference's .pt files are not in its repository), so
batches are generated: iid-uniform sequences over the 20 residues, backbone torsions from a
helix/sheet mixture, omega ~ pi, bond angles ~ N(1.94 / 2.03 / 2.13, 0.03), chi ~ U(-pi, pi).
Truth coordinates are built from the truth angles by whatever `build_coords` callable the
caller passes (the HIP NeRF in bench.py, the CPU oracle in tests), then the atom slots a
residue type does not own are set to NaN, as in the reference's data files
(protein/structure_utils.py:229-231).
"""
import numpy as np
import torch

N_SC = (1, 2, 4, 5, 7, 0, 6, 4, 5, 4, 4, 4, 3, 5, 7, 2, 3, 3, 10, 8)
PAD_ID = 20
DEFAULT_SEED = 11731          # reference default --seed (train.py:451)


def sample_angles(rng, L):
    a = np.zeros((L, 12), np.float64)
    helix = rng.random(L) < 0.5
    a[:, 0] = np.where(helix, -1.0, -2.1) + rng.normal(0, 0.3, L)
    a[:, 1] = np.where(helix, -0.8, 2.4) + rng.normal(0, 0.3, L)
    om = np.pi + rng.normal(0, 0.05, L)
    a[:, 2] = (om + np.pi) % (2 * np.pi) - np.pi
    a[:, 3] = rng.normal(1.94, 0.03, L)
    a[:, 4] = rng.normal(2.03, 0.03, L)
    a[:, 5] = rng.normal(2.13, 0.03, L)
    a[:, 6:] = rng.uniform(-np.pi, np.pi, (L, 6))
    return a.astype(np.float32)


def slot_mask(seq):
    """[B,L] ids -> bool [B, L*14]: True where the residue type owns the atom slot."""
    nsc = torch.tensor(N_SC + (0, 0), dtype=torch.int64)           # pad / unknown own nothing
    owned = 4 + nsc[seq.clamp(max=21).cpu()]
    owned = torch.where(seq.cpu() == PAD_ID, torch.zeros_like(owned), owned)
    slots = torch.arange(14)[None, None, :]
    return (slots < owned[:, :, None]).reshape(seq.shape[0], -1)


def make_batch(lens, L_pad=None, seed=DEFAULT_SEED, build_coords=None, frac_missing=0.0, noise=0.25):
    """Returns dict(seq [B,L] i64, true_ang [B,L,24] (cos,sin; NaN = unused chi; 0 = padding),
    true_ang_rad [B,L,12], start_ang_rad [B,L,12] (truth + noise, a plausible 'prediction'),
    true_crd [B,L*14,3] with NaN for absent atoms, or None when build_coords is None)."""
    rng = np.random.default_rng(seed)
    B = len(lens)
    L_pad = L_pad or max(lens)
    seq = np.full((B, L_pad), PAD_ID, np.int64)
    rad = np.zeros((B, L_pad, 12), np.float32)
    start = np.zeros((B, L_pad, 12), np.float32)
    sincos = np.zeros((B, L_pad, 24), np.float32)
    for b, L in enumerate(lens):
        s = rng.integers(0, 20, L)
        seq[b, :L] = s
        a = sample_angles(rng, L)
        rad[b, :L] = a
        st = a + rng.normal(0, noise, (L, 12)).astype(np.float32)
        start[b, :L] = (st + np.pi) % (2 * np.pi) - np.pi
        sc = np.stack([np.cos(a), np.sin(a)], -1).reshape(L, 24)
        for i, r in enumerate(s):
            sc[i, 12 + 2 * min(N_SC[int(r)], 6):] = np.nan
        sincos[b, :L] = sc
    out = dict(seq=torch.from_numpy(seq), true_ang=torch.from_numpy(sincos), true_ang_rad=torch.from_numpy(rad),
               start_ang_rad=torch.from_numpy(start), true_crd=None, lens=list(lens))
    if build_coords is not None:
        crd = build_coords(out["true_ang_rad"], out["seq"]).detach().float().cpu()      # [B, L*14, 3]
        own = slot_mask(out["seq"])
        if frac_missing:
            miss = torch.from_numpy(rng.random((B, L_pad)) < frac_missing)
            own = own & ~miss.repeat_interleave(14, dim=1)
        crd = torch.where(own[:, :, None], crd, torch.full_like(crd, float("nan")))
        pad = (out["seq"] == PAD_ID).repeat_interleave(14, dim=1)
        crd = torch.where(pad[:, :, None], torch.zeros_like(crd), crd)                  # collate pads with zeros
        out["true_crd"] = crd
    return out


def angle_means(true_ang):
    """nanmean over residues of the (cos,sin) truth, as scripts/proteinnet2pytorch.py:253-257 stores it."""
    flat = true_ang.reshape(-1, true_ang.shape[-1]).numpy()
    rows = (flat != 0).any(1)
    return np.nanmean(flat[rows], axis=0)
