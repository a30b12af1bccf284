"""PDB text writer (protein_transformer_amd.protein.PDB_Creator) against the text written by the reference's
PDB_Creator.save_pdb for the same coordinates (golden G11, tests/golden/make_golden.py::g11)."""
import numpy as np
import pytest


def test_pdb_text_matches_reference(golden, tmp_path):
    from protein_transformer_amd.protein.PDB_Creator import PDB_Creator
    g = golden("g11_pdb")
    for i in range(3):
        seq, crd, want = str(g[f"seq{i}"]), g[f"crd{i}"], str(g[f"pdb{i}"])
        path = tmp_path / f"p{i}.pdb"
        PDB_Creator(crd, seq=seq).save_pdb(str(path), title=f"golden {i}")
        assert path.read_text() == want
    # the missing atom of protein 1 is skipped, serial numbers stay consecutive, residue numbers are unaffected
    lines = str(g["pdb1"]).split("\n")
    serials = [int(l[6:11]) for l in lines if l.startswith("ATOM")]
    assert serials == list(range(1, len(serials) + 1))
    assert not any(l.startswith("ATOM") and l[12:16].strip() == "C" and int(l[22:26]) == 2 for l in lines)


def test_mapping_constructor_and_errors():
    from protein_transformer_amd.protein.PDB_Creator import ATOM_MAP_14, PDB_Creator
    crd = np.arange(2 * 14 * 3, dtype=np.float32).reshape(28, 3) + 1
    by_seq = PDB_Creator(crd, seq="AG").to_string("t")
    by_map = PDB_Creator(crd, mapping=[("A", ATOM_MAP_14["A"]), ("G", ATOM_MAP_14["G"])]).to_string("t")
    assert by_seq == by_map
    assert by_seq.count("ATOM") == 5 + 4                      # ALA: N CA C O CB, GLY: N CA C O
    assert all(len(v) == 14 for v in ATOM_MAP_14.values()) and len(ATOM_MAP_14) == 20
    with pytest.raises(Exception):
        PDB_Creator(crd)                                         # neither seq nor mapping
    with pytest.raises(AssertionError):
        PDB_Creator(crd, seq="AGA")                              # length mismatch
    with pytest.raises(NotImplementedError):
        PDB_Creator(crd, seq="AG").save_gltf("x.gltf")


def test_log_structure_writes_pred_and_true(golden, tmp_path):
    import types

    import torch
    from protein_transformer_amd.log import log_structure
    from protein_transformer_amd.protein.Sequence import VOCAB
    g = golden("g11_pdb")
    seq, crd = str(g["seq1"]), torch.tensor(g["crd1"])
    ids = torch.tensor(VOCAB.str2ints(seq, add_sos_eos=False))
    args = types.SimpleNamespace(structure_dir=str(tmp_path))
    true = torch.cat([crd, torch.zeros(28, 3)])                  # batch padding rows behind the protein
    pred_path, true_path = log_structure(args, crd.nan_to_num(1.0), true, ids, 7)
    assert pred_path.endswith("train/00007_pred.pdb")
    text = open(true_path).read()
    assert text.startswith("REMARK  true\n") and text.endswith("TER\nEND          \n")
    assert text.split("\n", 1)[1] == str(g["pdb1"]).split("\n", 1)[1]   # NaN atom -> zero row -> skipped, as upstream
    assert open(pred_path).read().count("ATOM") == text.count("ATOM") + 1
