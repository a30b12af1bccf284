"""`--synthetic B,L[,n_batches]`: a generated dataset in the reference's on-disk dictionary format
(scripts/proteinnet2pytorch.py:222-249), truth coordinates built by the HIP NeRF."""
import numpy as np
import torch

from . import synthetic
from .dataset import VALID_SPLITS
from .protein.Sequence import VOCAB
from .protein.Structure import nerf_forward


def make_synthetic_dataset(spec, seed, device):
    parts = [int(x) for x in spec.split(",")]
    B, L = parts[0], parts[1]
    n_batches = parts[2] if len(parts) > 2 else 8

    def split(n, s):
        build = lambda ang, seq: nerf_forward(ang.to(device), seq.to(device))[0]      # noqa: E731
        batch = synthetic.make_batch([L] * n, seed=s, build_coords=build)
        seqs = [VOCAB.ints2str(row.tolist()) for row in batch["seq"]]
        angs = [a.double().numpy() for a in batch["true_ang"]]
        crds = [c.double().numpy() for c in batch["true_crd"]]
        return {"seq": seqs, "ang": angs, "crd": crds, "ids": [f"SYN{s}_{i}" for i in range(n)]}, batch

    train, tb = split(B * n_batches, seed)
    data = {"train": train, "test": split(B, seed + 1)[0], "date": "synthetic", "description": {"spec": spec},
            "settings": {"max_len": L, "pad_char": np.nan, "angle_means": synthetic.angle_means(tb["true_ang"])}}
    for k, v in enumerate(VALID_SPLITS):
        data[f"valid-{v}"] = split(max(2, B // 4), seed + 2 + k)[0]
    return data
