"""Oracle: NeRF atom placement and the all-atom structure build (CPU, fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, op for op, the arithmetic of
  * `nerf`                      /root/reference/protein_transformer/protein/Structure.py:23-65
  * `StructureBuilder.build`    .../protein/StructureBuilder.py:71-92
  * `ResidueBuilder.build_bb`   .../protein/StructureBuilder.py:147-179
  * `ResidueBuilder.init_bb`    .../protein/StructureBuilder.py:181-191
  * `ResidueBuilder.build_sc`   .../protein/StructureBuilder.py:193-231
  * `stack_coords`              .../protein/StructureBuilder.py:233-236
  * the AMBER ff14SB build constants of .../protein/SidechainBuildInfo.py:1-585
as a flat table-driven program instead of the reference's builder objects.

Atom slots per residue: 0 N, 1 CA, 2 C, 3 O, 4.. side chain, zero padded to 14.
Angle columns per residue: 0 phi, 1 psi, 2 omega, 3 N-CA-C, 4 CA-C-N, 5 C-N-CA,
6.. chi.
"""
import math

import torch

NUM_ANGLES = 12
NUM_SLOTS = 14
SC_ANGLE0 = 6
PAD_ID = 20

# backbone constants (SidechainBuildInfo.py:576-585)
BL_N_CA = 1.442
BL_CA_C = 1.498
BL_C_N = 1.379
BL_C_O = 1.229
BA_CA_C_O = 2.0944

_PI = 3.141592653589793
_CB = (1.526, 1.9146261894377796)
_T = 1.911135530933791      # tetrahedral-ish CT angle used all over ff14SB
_R = 2.0943951023931953     # 2*pi/3 ring angle

# Side-chain programs, one tuple per atom: (bond, angle, torsion, parents)
#   torsion: "p" predicted chi (column 6+k), "i" previous torsion - pi,
#            or a float constant.
#   parents: slots (a, b, c) of the three atoms the new atom hangs off; the
#            first atom (CB) has parents None = (C of previous residue, N, CA)
#            or, for the first residue, (N of next residue, C, CA).
SC_PROGRAM = {
    0:  [(*_CB, "p", None)],                                                   # A
    1:  [(*_CB, "p", None), (1.81, 1.8954275676658419, "p", (0, 1, 4))],       # C
    2:  [(*_CB, "p", None), (1.522, 1.9390607989657, "p", (0, 1, 4)),          # D
         (1.25, 2.0420352248333655, "p", (1, 4, 5)),
         (1.25, 2.0420352248333655, "i", (1, 4, 5))],
    3:  [(*_CB, "p", None), (1.526, _T, "p", (0, 1, 4)),                       # E
         (1.522, 1.9390607989657, "p", (1, 4, 5)),
         (1.25, 2.0420352248333655, "p", (4, 5, 6)),
         (1.25, 2.0420352248333655, "i", (4, 5, 6))],
    4:  [(*_CB, "p", None), (1.51, 1.9896753472735358, "p", (0, 1, 4)),        # F
         (1.4, _R, "p", (1, 4, 5)), (1.4, _R, _PI, (4, 5, 6)),
         (1.4, _R, 0.0, (5, 6, 7)), (1.4, _R, 0.0, (6, 7, 8)),
         (1.4, _R, 0.0, (7, 8, 9))],
    5:  [],                                                                    # G
    6:  [(*_CB, "p", None), (1.504, 1.9739673840055867, "p", (0, 1, 4)),       # H
         (1.385, _R, "p", (1, 4, 5)),
         (1.343, 1.8849555921538759, _PI, (4, 5, 6)),
         (1.335, 1.8849555921538759, 0.0, (5, 6, 7)),
         (1.394, 1.8849555921538759, 0.0, (6, 7, 8))],
    7:  [(*_CB, "p", None), (1.526, _T, "p", (0, 1, 4)),                       # I
         (1.526, _T, "p", (1, 4, 5)), (1.526, _T, "p", (0, 1, 4))],
    8:  [(*_CB, "p", None), (1.526, _T, "p", (0, 1, 4)),                       # K
         (1.526, _T, "p", (1, 4, 5)), (1.526, _T, "p", (4, 5, 6)),
         (1.471, 1.9408061282176945, "p", (5, 6, 7))],
    9:  [(*_CB, "p", None), (1.526, _T, "p", (0, 1, 4)),                       # L
         (1.526, _T, "p", (1, 4, 5)), (1.526, _T, "p", (1, 4, 5))],
    10: [(*_CB, "p", None), (1.526, _T, "p", (0, 1, 4)),                       # M
         (1.81, 2.0018926520374962, "p", (1, 4, 5)),
         (1.81, 1.726130630222392, "p", (4, 5, 6))],
    11: [(*_CB, "p", None), (1.522, 1.9390607989657, "p", (0, 1, 4)),          # N
         (1.229, 2.101376419401173, "p", (1, 4, 5)),
         (1.335, 2.035053907825388, "i", (1, 4, 5))],
    12: [(*_CB, "p", None), (1.526, _T, "p", (0, 1, 4)),                       # P
         (1.526, _T, "p", (1, 4, 5))],
    13: [(*_CB, "p", None), (1.526, _T, "p", (0, 1, 4)),                       # Q
         (1.522, 1.9390607989657, "p", (1, 4, 5)),
         (1.229, 2.101376419401173, "p", (4, 5, 6)),
         (1.335, 2.035053907825388, "i", (4, 5, 6))],
    14: [(*_CB, "p", None), (1.526, _T, "p", (0, 1, 4)),                       # R
         (1.526, _T, "p", (1, 4, 5)),
         (1.463, 1.9408061282176945, "p", (4, 5, 6)),
         (1.34, 2.150245638457014, "p", (5, 6, 7)),
         (1.34, _R, "p", (6, 7, 8)), (1.34, _R, "i", (6, 7, 8))],
    15: [(*_CB, "p", None), (1.41, _T, "p", (0, 1, 4))],                       # S
    16: [(*_CB, "p", None), (1.41, _T, "p", (0, 1, 4)),                        # T
         (1.526, _T, "p", (0, 1, 4))],
    17: [(*_CB, "p", None), (1.526, _T, "p", (0, 1, 4)),                       # V
         (1.526, _T, "p", (0, 1, 4))],
    18: [(*_CB, "p", None), (1.495, 2.0176006153054447, "p", (0, 1, 4)),       # W
         (1.352, 2.181661564992912, "p", (1, 4, 5)),
         (1.381, 1.8971728969178363, _PI, (4, 5, 6)),
         (1.38, 1.9477874452256716, 0.0, (5, 6, 7)),
         (1.4, 2.3177972466484698, _PI, (6, 7, 8)),
         (1.4, _R, _PI, (7, 8, 9)), (1.4, _R, 0.0, (8, 9, 10)),
         (1.4, _R, 0.0, (9, 10, 11)), (1.404, _R, 0.0, (10, 11, 12))],
    19: [(*_CB, "p", None), (1.51, 1.9896753472735358, "p", (0, 1, 4)),        # Y
         (1.4, _R, "p", (1, 4, 5)), (1.4, _R, _PI, (4, 5, 6)),
         (1.409, _R, 0.0, (5, 6, 7)), (1.364, _R, _PI, (6, 7, 8)),
         (1.409, _R, 0.0, (6, 7, 8)), (1.4, _R, 0.0, (7, 8, 10))],
}

AA_LETTERS = "ACDEFGHIKLMNPQRSTVWY"


def seq_to_ids(s):
    """1-letter string -> list of ids (Sequence.py:81-89: A..Y alphabetical -> 0..19)."""
    return [AA_LETTERS.index(ch) for ch in s]


def _unit(v):
    # torch.nn.functional.normalize(v, dim=0): v / max(||v||, 1e-12)   (Structure.py:44-45,50)
    return torch.nn.functional.normalize(v, dim=0)


def nerf(a, b, c, l, theta, chi):
    """Place atom d given a, b, c, bond length l, bond angle theta, torsion chi.

    Structure.py:23-65.  l may be a Python float (backbone) or an fp32 tensor
    (side chain); theta/chi are fp32 tensors.
    """
    assert -math.pi <= float(theta.detach() if torch.is_tensor(theta) else theta) <= math.pi, "theta must be in [-pi, pi]"   # Structure.py:42
    w_hat = _unit(b - a)
    x_hat = _unit(c - b)
    n = torch.linalg.cross(w_hat, x_hat)
    z_hat = _unit(n)
    y_hat = torch.linalg.cross(z_hat, x_hat)
    m = torch.stack([x_hat, y_hat, z_hat], dim=1)
    d = torch.stack([torch.squeeze(-l * torch.cos(theta)),
                     torch.squeeze(l * torch.sin(theta) * torch.cos(chi)),
                     torch.squeeze(l * torch.sin(theta) * torch.sin(chi))])
    d = d.unsqueeze(1).to(torch.float32)
    return (c + torch.mm(m, d).squeeze()).squeeze()


def _backbone_first(ang0):
    """init_bb (StructureBuilder.py:181-191): N, CA fixed; C detached from the graph."""
    n = torch.tensor([0, 0, 0.001])
    ca = n + torch.tensor([BL_N_CA, 0, 0])
    cx = (torch.cos(math.pi - ang0[3]) * BL_CA_C).detach()     # re-wrapped by the reference => detached
    cy = (torch.sin(math.pi - ang0[3]) * BL_CA_C).detach()
    c = ca + torch.tensor([cx, cy, 0], dtype=torch.float32)
    o = nerf(n, ca, c, torch.tensor(BL_C_O), torch.tensor(BA_CA_C_O), ang0[1] - math.pi)
    return [n, ca, c, o]


def _backbone_next(prev_bb, prev_ang, ang):
    """build_bb for residue i>=1 (StructureBuilder.py:147-179)."""
    pn, pca, pc = prev_bb[0], prev_bb[1], prev_bb[2]
    n = nerf(pn, pca, pc, BL_C_N, prev_ang[4], prev_ang[1])
    ca = nerf(pca, pc, n, BL_N_CA, prev_ang[5], prev_ang[2])
    c = nerf(pc, n, ca, BL_CA_C, ang[3], ang[0])
    o = nerf(n, ca, c, BL_C_O, torch.tensor(BA_CA_C_O), ang[1] - math.pi)
    return [n, ca, c, o]


def _sidechain(res_id, ang, bb, c_prev=None, n_next=None):
    """build_sc (StructureBuilder.py:193-231). Returns list of side-chain atoms."""
    slots = {0: bb[0], 1: bb[1], 2: bb[2]}
    out = []
    last_torsion = None
    for k, (bond, angle, tors, parents) in enumerate(SC_PROGRAM[int(res_id)]):
        if k == 0:
            if n_next is not None:
                a, b, c = n_next, slots[2], slots[1]
            else:
                a, b, c = c_prev, slots[0], slots[1]
        else:
            a, b, c = (slots[p] for p in parents)
        if tors == "p":
            chi = ang[SC_ANGLE0 + k]
        elif tors == "i":
            chi = last_torsion - math.pi
        else:
            chi = torch.tensor(tors, dtype=torch.float32)
        pt = nerf(a, b, c, torch.tensor(bond, dtype=torch.float32),
                  torch.tensor(angle, dtype=torch.float32), chi)
        slots[4 + k] = pt
        out.append(pt)
        last_torsion = chi
    return out


def generate_coords(angles, input_seq, device=None):
    """[L,12] radians + [L] residue ids -> [L*14, 3] fp32 coordinates.

    Structure.py:12-20 -> StructureBuilder.build (StructureBuilder.py:71-92).
    Residue 0's side chain is built after residue 1's backbone because its CB
    hangs off N of residue 1 (StructureBuilder.py:55-69).
    """
    if isinstance(input_seq, str):
        input_seq = torch.tensor(seq_to_ids(input_seq))
    L = len(input_seq)
    if L < 2:
        raise StopIteration("structure build needs at least two residues")   # StructureBuilder.py:58-59
    for r in input_seq:
        if not 0 <= int(r) < 20:
            raise KeyError(int(r))                                             # Sequence.py:50-51
    pad = torch.zeros(3)
    bbs = [None] * L
    bbs[0] = _backbone_first(angles[0])
    for i in range(1, L):
        bbs[i] = _backbone_next(bbs[i - 1], angles[i - 1], angles[i])
    rows = []
    for i in range(L):
        if i == 0:
            sc = _sidechain(input_seq[0], angles[0], bbs[0], n_next=bbs[1][0])
        else:
            sc = _sidechain(input_seq[i], angles[i], bbs[i], c_prev=bbs[i - 1][2])
        atoms = bbs[i] + sc
        rows += atoms + [pad] * (NUM_SLOTS - len(atoms))
    return torch.stack(rows)


def n_atoms_of(res_id):
    return 4 + len(SC_PROGRAM[int(res_id)])
