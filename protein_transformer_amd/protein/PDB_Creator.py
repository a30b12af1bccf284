"""PDB text writer for predicted all-atom structures (host side, no GPU work).

Mirrors the reference's `PDB_Creator(coords, seq=None, mapping=None, atoms_per_res=14)` with `save_pdb(path, title)`
(/root/reference/protein_transformer/protein/PDB_Creator.py:17-172) and its `ATOM_MAP_14` (:227-231), so that
`log.py`-style structure dumps work without PyMOL / ProDy / wandb.  The glTF / PNG
exports of the reference (`save_gltf`, `save_gltfs`, :174-212) need PyMOL and are out of scope; they raise.

Behaviour kept from the reference:
  * one ATOM record per atom in slot order N, CA, C, O, side chain; atoms named "PAD", with a NaN coordinate or whose
    coordinates sum to exactly 0 are skipped (:112-123), and atom serial numbers only count written atoms;
  * residue numbers count every residue, written or not (:131-137);
  * fixed-column record "{:6s}{:5d} {:^4s}{:1s}{:3s} {:1s}{:4d}{:1s}   {:8.3f}{:8.3f}{:8.3f}{:6.2f}{:6.2f}
    {:>2s}{:2s}" with empty chain / altloc / insertion code, occupancy 1, temperature factor 0 and the first letter of
    the atom name as the element symbol (:55-63,84-102);
  * header "REMARK  <title>", footer "TER\\nEND          \\n", lines joined by "\\n" (:139-172).
"""
import numpy as np

from .Sequence import ONE_TO_THREE_LETTER_MAP

NUM_PREDICTED_COORDS = 14

# heavy side-chain atoms in build order (the slot order of generate_coords / ptamd_nerf_fwd)
_SIDECHAIN_ATOMS = {
    "ALA": ["CB"], "ARG": ["CB", "CG", "CD", "NE", "CZ", "NH1", "NH2"], "ASN": ["CB", "CG", "OD1", "ND2"],
    "ASP": ["CB", "CG", "OD1", "OD2"], "CYS": ["CB", "SG"], "GLN": ["CB", "CG", "CD", "OE1", "NE2"],
    "GLU": ["CB", "CG", "CD", "OE1", "OE2"], "GLY": [], "HIS": ["CB", "CG", "ND1", "CE1", "NE2", "CD2"],
    "ILE": ["CB", "CG1", "CD1", "CG2"], "LEU": ["CB", "CG", "CD1", "CD2"], "LYS": ["CB", "CG", "CD", "CE", "NZ"],
    "MET": ["CB", "CG", "SD", "CE"], "PHE": ["CB", "CG", "CD1", "CE1", "CZ", "CE2", "CD2"], "PRO": ["CB", "CG", "CD"],
    "SER": ["CB", "OG"], "THR": ["CB", "OG1", "CG2"],
    "TRP": ["CB", "CG", "CD1", "NE1", "CE2", "CZ2", "CH2", "CZ3", "CE3", "CD2"],
    "TYR": ["CB", "CG", "CD1", "CE1", "CZ", "OH", "CE2", "CD2"], "VAL": ["CB", "CG1", "CG2"],
}

ATOM_MAP_14 = {}
for _one, _three in ONE_TO_THREE_LETTER_MAP.items():
    _names = ["N", "CA", "C", "O"] + _SIDECHAIN_ATOMS[_three]
    ATOM_MAP_14[_one] = _names + ["PAD"] * (NUM_PREDICTED_COORDS - len(_names))

_ATOM_FORMAT = ("{:6s}{:5d} {:^4s}{:1s}{:3s} {:1s}{:4d}{:1s}   {:8.3f}{:8.3f}{:8.3f}{:6.2f}{:6.2f}          "
                "{:>2s}{:2s}")


class PDB_Creator(object):
    """Turns an (L * atoms_per_res) x 3 coordinate array plus a sequence (or a residue -> atom-name mapping) into
    PDB text."""

    def __init__(self, coords, seq=None, mapping=None, atoms_per_res=NUM_PREDICTED_COORDS):
        if hasattr(coords, "detach"):
            coords = coords.detach().cpu().numpy()
        self.coords = np.asarray(coords)
        if seq and not mapping:
            assert len(seq) == self.coords.shape[0] / atoms_per_res, \
                "The sequence length must match the coordinate length and contain 1 letter AA codes." + \
                str(self.coords.shape[0] / atoms_per_res) + " " + str(len(seq))
            self.seq = seq
            self.mapping = [(res, ATOM_MAP_14[res]) for res in seq]
        elif not seq and not mapping:
            raise Exception("Please provide a seq or a mapping.")
        elif mapping and not seq:
            self.mapping = mapping
            self.seq = "".join(m[0] for m in mapping)
        else:
            self.seq, self.mapping = seq, mapping
        assert type(self.mapping[0][0]) == str and len(self.mapping[0][0]) == 1, \
            "1 letter AA codes must be used in the mapping."
        self.atoms_per_res = atoms_per_res
        assert self.coords.shape[0] % self.atoms_per_res == 0, \
            f"Coords is not divisible by {atoms_per_res}. {self.coords.shape}"
        self.lines = []

    def _atom_line(self, serial, res_nbr, res_name, atom_name, xyz):
        return _ATOM_FORMAT.format("ATOM", serial, atom_name, "", ONE_TO_THREE_LETTER_MAP[res_name], "", res_nbr, "",
                                   xyz[0], xyz[1], xyz[2], 1, 0, atom_name[0], "")

    def _get_lines_for_protein(self):
        self.lines = []
        serial = 1
        n = self.atoms_per_res
        for res_idx, (res_name, atom_names) in enumerate(self.mapping):
            res_coords = self.coords[res_idx * n:(res_idx + 1) * n]
            for atom_name, xyz in zip(atom_names, res_coords):
                if atom_name == "PAD" or np.isnan(xyz).sum() > 0 or xyz.sum() == 0:
                    continue
                self.lines.append(self._atom_line(serial, res_idx + 1, res_name, atom_name, xyz))
                serial += 1
        return self.lines

    def to_string(self, title="test"):
        self._get_lines_for_protein()
        self.lines = [f"REMARK  {title}"] + self.lines + ["TER\nEND          \n"]
        return "\n".join(self.lines)

    def save_pdb(self, path, title="test"):
        with open(path, "w") as outfile:
            outfile.write(self.to_string(title))

    def save_gltf(self, *args, **kwargs):
        raise NotImplementedError("glTF export needs PyMOL (PDB_Creator.py:174-184 of the reference): out of scope")

    save_gltfs = save_gltf
