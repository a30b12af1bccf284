"""Host data path vs the golden vectors captured from the reference's dataset.py (G9), plus the reference's own
tests/test_datasets.py:20-31 bin-bookkeeping case.  CPU only."""
import numpy as np
import torch

from protein_transformer_amd import dataset as D
from protein_transformer_amd.protein.Sequence import VOCAB


def _g9(golden):
    g = golden("g9_dataset")
    n = int(g["n"])
    return g, [str(g[f"seq{i}"]) for i in range(n)], [g[f"ang{i}"] for i in range(n)], [g[f"crd{i}"] for i in range(n)]


def test_binned_dataset_golden(golden):
    g, seqs, angs, crds = _g9(golden)
    ds = D.BinnedProteinDataset(seqs=seqs, angs=angs, crds=crds, add_sos_eos=False, skip_missing_residues=False, bins="auto")
    assert ds.lens == list(g["lens"])
    assert np.array_equal(ds.hist_bins, g["hist_bins"]) and np.array_equal(ds.hist_counts, g["hist_counts"])
    assert np.allclose(ds.bin_probs, g["bin_probs"])
    assert sorted(ds.bin_map) == list(g["bin_map_keys"])
    for k, v in ds.bin_map.items():
        assert v == list(g[f"bin_map_{k}"])


def test_collate_golden(golden):
    g, seqs, angs, crds = _g9(golden)
    ds = D.BinnedProteinDataset(seqs=seqs, angs=angs, crds=crds, add_sos_eos=False, skip_missing_residues=False)
    s, a, c = D.paired_collate_fn([ds[i] for i in (1, 3, 0)])
    assert s.dtype == torch.int64 and a.dtype == torch.float32 and c.dtype == torch.float32
    assert np.array_equal(s.numpy(), g["collate_seq"])
    assert np.array_equal(a.numpy(), g["collate_ang"]) and np.array_equal(c.numpy(), g["collate_crd"])
    assert int(s[2, 5]) == VOCAB.pad_id                      # shortest sequence is padded with id 20
    # truncation to max_seq_len residues / 14x atoms
    # the batch is packed: three views of one buffer (one upload, dataset.DevicePrefetcher), the plain tensors on request
    assert D.packed_base((s, a, c)) is not None and s.dtype == torch.int64 and a.dtype == c.dtype == torch.float32
    plain = D.make_paired_collate_fn(D.MAX_SEQ_LEN, packed=False)([ds[i] for i in (1, 3, 0)])
    assert D.packed_base(plain) is None and all(torch.equal(x, y) for x, y in zip(plain, (s, a, c)))
    s2, a2, c2 = D.make_paired_collate_fn(8)([ds[i] for i in (1, 3, 0)])
    assert s2.shape == (3, 8) and a2.shape == (3, 8, 24) and c2.shape == (3, 8 * 14, 3)


def test_sampler_golden(golden):
    g, seqs, angs, crds = _g9(golden)
    ds = D.BinnedProteinDataset(seqs=seqs, angs=angs, crds=crds, add_sos_eos=False, skip_missing_residues=False)
    for cpu_opt in (False, True):
        sampler = D.SimilarLengthBatchSampler(ds, 4, dynamic_batch=200, optimize_batch_for_cpus=cpu_opt)
        sampler.cpu_count = 2
        assert len(sampler) == int(g[f"sampler_len_cpuopt{int(cpu_opt)}"])
        np.random.seed(7)
        assert [len(b) for b in sampler] == list(g[f"sampler_sizes_cpuopt{int(cpu_opt)}"])
    np.random.seed(7)
    sampler = D.SimilarLengthBatchSampler(ds, 4, dynamic_batch=200, optimize_batch_for_cpus=False)
    assert np.array_equal(next(iter(sampler)), g["sampler_first_batch"])     # same draws from the same RNG state
    pds = D.ProteinDataset(seqs=seqs, angs=angs, crds=crds, add_sos_eos=False, skip_missing_residues=False)
    assert [len(pds[i][0]) for i in range(len(pds))] == list(g["pds_order_lens"])


def test_binned_dataset_reference_case():
    # /root/reference/protein_transformer/tests/test_datasets.py:20-31
    seqs = ["A" * 5, "A" * 10, "A" * 10, "A" * 21]
    angs = [np.random.random((len(s), 24)) for s in seqs]
    crds = [np.random.random((len(s) * 14, 3)) for s in seqs]
    ds = D.BinnedProteinDataset(seqs=seqs, angs=angs, crds=crds, add_sos_eos=False, skip_missing_residues=False, bins=4)
    assert len(ds) == 4
    assert list(ds.hist_counts) == [1, 2, 0, 1]
    assert ds.bin_map == {0: [0], 1: [1, 2], 3: [3]}


def test_skip_missing_residues():
    seqs = ["AC", "ACD"]
    angs = [np.ones((2, 24)), np.ones((3, 24))]
    angs[1][1] = np.nan
    crds = [np.ones((28, 3)), np.ones((42, 3))]
    assert len(D.ProteinDataset(seqs, angs, crds, add_sos_eos=False, skip_missing_residues=True)) == 1
    assert len(D.ProteinDataset(seqs, angs, crds, add_sos_eos=False, skip_missing_residues=False)) == 2
