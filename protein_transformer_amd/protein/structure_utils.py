"""Host-side selection helpers of the reference's `protein/structure_utils.py` that callers of the loss path import.

Reference: /root/reference/protein_transformer/protein/structure_utils.py:19-41.  Only the two coordinate selectors are on
the drop-in surface (losses.py:12,70-71,83-84 import and call `get_backbone_from_full_coords`); the ProDy / PDB parsing
half of that file is dataset construction (out of scope).  Inside the step nothing calls these: the
dRMSD kernels pick the backbone slots themselves while they compact the atoms (csrc/drmsd.hip, "backbone first").  They
exist so that `import protein_transformer_amd as protein_transformer` works for code that slices coordinates itself.

Row selection only - no arithmetic - on whatever the caller hands over (numpy array or torch tensor, on any device), with
or without a leading batch dimension, exactly like the reference (whose `np.bool` no longer exists in numpy >= 1.24).
"""
import numpy as np

from .Structure import NUM_PREDICTED_COORDS


def _slot_mask(n_rows, invert):
    """True for the rows to keep in a [n_rows, 3] coordinate block whose rows cycle through the 14 atom slots of a residue
    (N, CA, C first: structure_utils.py:26)."""
    mask = np.array([True, True, True] + [False] * (NUM_PREDICTED_COORDS - 3), dtype=bool)
    if invert:
        mask = np.invert(mask)
    return np.tile(mask, n_rows // NUM_PREDICTED_COORDS)


def get_backbone_from_full_coords(crds, invert=False):
    """Coordinates [L * 14, 3] or [B, L * 14, 3] -> the backbone atoms (slots 0-2 of every residue) [L * 3, 3] /
    [B, L * 3, 3]; `invert` keeps the other 11 slots instead (structure_utils.py:19-32)."""
    if len(crds.shape) == 2:
        mask = _slot_mask(crds.shape[0], invert)
        if not isinstance(crds, np.ndarray):      # torch tensor: a boolean mask on its device
            import torch
            return crds[torch.as_tensor(mask, device=crds.device), :]
        return crds[mask, :]
    mask = _slot_mask(crds.shape[1], invert)
    if not isinstance(crds, np.ndarray):
        import torch
        return crds[:, torch.as_tensor(mask, device=crds.device), :]
    return crds[:, mask, :]


def get_sidechain_from_full_coords(crds):
    """The complement of `get_backbone_from_full_coords` (structure_utils.py:35-41)."""
    return get_backbone_from_full_coords(crds, invert=True)
