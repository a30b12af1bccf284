"""Data-parallel host logic on CPU with the gloo backend, world_size 2: batch sharding and the SUM
all-reduce of the flat gradient buffer reproduce the single-process gradient of the whole batch
(the reference back-propagates the SUM over proteins, losses.py:166-167 / SURVEY.md A-7)."""
import os
import socket
import types

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import encoder as oenc, losses as olosses
    from protein_transformer_amd import dp, synthetic
    torch.set_num_threads(1)
    dp.init_from_env(backend="gloo")
    assert dp.world_size() == world and dp.rank() == rank
    from oracle import geometry
    lens = [9, 7, 8, 6, 5]                       # 5 proteins over 2 ranks: shards of 3 and 2
    build = lambda ang, seq: torch.stack([                                     # noqa: E731
        torch.cat([geometry.generate_coords(ang[b, :n], seq[b, :n]), torch.zeros((seq.shape[1] - n) * 14, 3)])
        for b, n in enumerate(lens)])
    batch = synthetic.make_batch(lens, seed=4, build_coords=build)
    am = synthetic.angle_means(batch["true_ang"])
    params = oenc.init_params(1, 32, 64, 16, am, seed=2)
    params["output_projection.weight"].normal_(0, 0.05)
    names = [k for k in params if not k.endswith(".pe")]

    def flat_grad(seq, crd):
        leaf = {k: params[k].clone().requires_grad_() for k in names}
        pred = oenc.encoder_forward({**leaf, "encoder.positional_enc.pe": params["encoder.positional_enc.pe"]}, seq, 4)
        olosses.compute_batch_drmsd(pred, crd, seq, do_backward=True)
        return torch.cat([leaf[k].grad.reshape(-1) for k in names])

    seq, crd = dp.shard_batch(batch["seq"], batch["true_crd"])
    assert seq.shape[0] == (3 if rank == 0 else 2)
    model = types.SimpleNamespace(_flat=None, _flat_grad=flat_grad(seq, crd), grad_hook=None)
    model.flat_parameters = lambda: (model._flat, model._flat_grad)
    # overlapped path: reduce two slices as "backward" finishes them, then wait
    dp.attach(model)
    n = model._flat_grad.numel()
    model.grad_slices = lambda: [(0, n // 2), (n // 2, n - n // 2)]
    model.grad_hook(0, n // 2)
    model.grad_hook(n // 2, n - n // 2)
    dp.all_reduce_gradients(model)
    assert model._dp_pending == []
    if rank == 0:
        full = flat_grad(batch["seq"], batch["true_crd"])
        np.save(os.path.join(out_dir, "dp.npy"), model._flat_grad.numpy())
        np.save(os.path.join(out_dir, "full.npy"), full.numpy())
    # non-overlapped path gives the same thing
    model2 = types.SimpleNamespace(_flat=None, _flat_grad=flat_grad(seq, crd), grad_hook=None)
    model2.flat_parameters = lambda: (model2._flat, model2._flat_grad)
    dp.all_reduce_gradients(model2)
    assert torch.allclose(model2._flat_grad, model._flat_grad, rtol=1e-6, atol=1e-9)
    t = dp.all_reduce_sum_(torch.tensor([float(rank + 1)]))
    assert float(t) == 3.0
    # a rank with an EMPTY shard (fewer proteins than ranks) has run no backward: it walks the model's slice list with a
    # zero buffer and must end with rank 0's gradient
    g_one = flat_grad(batch["seq"][:1], batch["true_crd"][:1])
    model3 = types.SimpleNamespace(_flat=None, _flat_grad=g_one.clone() if rank == 0 else torch.zeros_like(g_one), grad_hook=None)
    model3.flat_parameters = lambda: (model3._flat, model3._flat_grad)
    model3.grad_slices = lambda: [(0, 100), (100, n - 100)]
    dp.attach(model3)
    if rank == 0:
        for off, cnt in model3.grad_slices():
            model3.grad_hook(off, cnt)
    dp.all_reduce_gradients(model3, empty=rank != 0)
    assert torch.equal(model3._flat_grad, g_one)
    # sharded loaders: the ranks' batches partition the global batches; dRMSD batch sizes are multiples of the rank count
    from protein_transformer_amd.dataset import prepare_dataloaders
    from protein_transformer_amd.protein.Sequence import VOCAB
    rng = np.random.default_rng(0)
    lens2 = sorted(int(x) for x in rng.integers(5, 60, 40))
    seqs = ["".join(VOCAB.int2char(int(i)) for i in rng.integers(0, 20, m)) for m in lens2]
    split = {"seq": seqs, "ang": [np.full((m, 24), float(m)) for m in lens2], "crd": [np.zeros((m * 14, 3)) for m in lens2]}
    data = {"train": split, "valid-10": split, "test": split}
    a = types.SimpleNamespace(batching_order="binned-random", loss="drmsd", add_sos_eos=False, skip_missing_res_train=False,
                              bins=4, batch_size=3, repeat_train=1, train_eval_downsample=0.5)
    tag = lambda t: [int(x) for x in t[:, 0, 0]] if t.shape[0] else []      # noqa: E731
    np.random.seed(11)
    tr, tre, val, te = prepare_dataloaders(data, a, 60, num_workers=0)
    got, widths = [], []
    for i, (s_, a_, c_) in enumerate(tr):
        got.append(tag(a_))                                 # the angle tensor carries the protein's length as a tag
        widths.append(int(s_.shape[1]) if s_.shape[0] else None)
        assert s_.shape[0] == a_.shape[0] == c_.shape[0] and (s_.shape[0] == 0 or s_.shape[1] >= max(got[-1]))
        assert s_.shape[0] == 0 or c_.shape[1] == 14 * s_.shape[1] == 14 * a_.shape[1]
    ev = [tag(a_) for _, a_, _ in te]
    out = [None, None]
    torch.distributed.all_gather_object(out, (got, ev, widths))
    # every shard is padded to the longest protein of the GLOBAL batch (conv-enc sees the same columns as single-process)
    for b0, b1, w0, w1 in zip(out[0][0], out[1][0], out[0][2], out[1][2]):
        assert {w for w in (w0, w1) if w is not None} == {max(b0 + b1)}
    for b0, b1 in zip(out[0][0], out[1][0]):
        assert (len(b0) + len(b1)) % 2 == 0 and abs(len(b0) - len(b1)) == 0          # multiples of the rank count, dealt evenly
        both = sorted(b0 + b1, reverse=True)
        assert abs(sum(b0) - sum(b1)) <= max(both)                                   # length-balanced
    assert len(out[0][0]) == len(out[1][0]) > 0
    flat_eval = sorted(x for r in (0, 1) for b in out[r][1] for x in b)
    assert flat_eval == sorted(lens2)                                               # evaluation covers every protein once
    # the self-check block of the bench line: who is in the job, and do the ranks hold the same parameters
    who = dp.describe()
    assert who["backend"] == "gloo" and who["world_size"] == world and who["ranks_ok"]
    assert sorted(r["rank"] for r in who["ranks_seen"]) == [0, 1] and len({r["pid"] for r in who["ranks_seen"]}) == 2
    same = types.SimpleNamespace(flat_parameters=lambda: (torch.arange(8, dtype=torch.float32) - 3.5, None))
    sp = dp.param_checksum_spread(same)
    assert sp["abs_sum_spread"] == 0.0 and sp["bit_hash_spread"] == 0 and sp["ranks"] == 2 and sp["abs_sum"] == 16.0
    differ = types.SimpleNamespace(flat_parameters=lambda: (torch.arange(8, dtype=torch.float32) + 1e-3 * rank, None))
    sp = dp.param_checksum_spread(differ)
    assert sp["abs_sum_spread"] > 0 and sp["bit_hash_spread"] != 0                  # one ulp on one rank is seen
    # single_process(): this rank acts as a world of one (no collective, hook detached) while the group stays up
    hooked = types.SimpleNamespace(grad_hook=lambda *a: 1 / 0)
    with dp.single_process(hooked):
        assert dp.world_size() == 1 and dp.rank() == 0 and hooked.grad_hook is None
        t1 = dp.all_reduce_sum_(torch.tensor([5.0]))
        assert float(t1) == 5.0
    assert dp.world_size() == world and dp.rank() == rank and hooked.grad_hook is not None
    dp.barrier()
    dp.shutdown()


def test_shard_bounds():
    from protein_transformer_amd.dp import shard_bounds
    assert [shard_bounds(32, 8, r) for r in range(8)] == [(4 * r, 4 * r + 4) for r in range(8)]
    assert [shard_bounds(5, 2, r) for r in range(2)] == [(0, 3), (3, 5)]
    assert [shard_bounds(3, 4, r) for r in range(4)] == [(0, 1), (1, 2), (2, 3), (3, 3)]


def test_shard_indices_serpentine():
    from protein_transformer_amd.dp import shard_indices
    lens = [50, 10, 40, 30, 20, 60, 5]
    parts = [shard_indices(lens, 3, r) for r in range(3)]
    assert sorted(i for p in parts for i in p) == list(range(7))                    # a partition
    assert [len(p) for p in parts] == [3, 2, 2]
    # sorted by length: 60 50 40 | 30 20 10 | 5 -> ranks 0 1 2 | 2 1 0 | 0
    assert [sorted(lens[i] for i in p) for p in parts] == [[5, 10, 60], [20, 50], [30, 40]]
    assert shard_indices(lens, 1, 0) == list(range(7))
    assert [shard_indices([7], 2, r) for r in range(2)] == [[0], []]                # fewer proteins than ranks
    assert shard_indices([], 4, 2) == []
    # equal lengths: ties broken by position, deterministic
    assert [shard_indices([9] * 8, 4, r) for r in range(4)] == [[0, 7], [1, 6], [2, 5], [3, 4]]


def test_sampler_batches_are_multiples_of_the_workers():
    """the reference rounds dRMSD batches down to a multiple of cpu_count() (dataset.py:218-220); on a
    many-core GPU host that collapsed every batch to one protein.  The workers of this path are the GPUs."""
    from protein_transformer_amd import dataset as D
    seqs = ["A" * n for n in (5, 10, 10, 21, 30, 30, 31, 64)]
    angs = [np.ones((len(s), 24)) for s in seqs]
    crds = [np.ones((len(s) * 14, 3)) for s in seqs]
    ds = D.BinnedProteinDataset(seqs=seqs, angs=angs, crds=crds, add_sos_eos=False, skip_missing_residues=False, bins=3)
    smp = D.SimilarLengthBatchSampler(ds, 4, dynamic_batch=256, optimize_batch_for_cpus=True)
    smp.cpu_count, smp.min_batch = 8, 8                                             # an 8-GPU job
    np.random.seed(0)
    sizes = {len(b) for b in smp}
    assert sizes and all(sz % 8 == 0 and sz >= 8 for sz in sizes)
    smp.cpu_count, smp.min_batch = 1, 1                                             # one GPU: the plain residue budget
    np.random.seed(0)
    assert {len(b) for b in smp} <= {int(256 / e) for e in ds.hist_bins}


def test_dp_sum_allreduce_equals_full_batch(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got, full = np.load(tmp_path / "dp.npy"), np.load(tmp_path / "full.npy")
    assert np.abs(got - full).max() <= 1e-5 * np.abs(full).max()


def _worker4(rank, world, port, out_dir):
    """world_size 4, shards of 2 / 1 / 1 / 0 proteins: every rank issues the same per-slice reductions in the same order
    (the rank with nothing walks the slice list with a zero buffer), the loss statistics come out as statistics of the
    global batch on every rank, and the comm meter counts the bytes it should."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import encoder as oenc, geometry, losses as olosses
    from protein_transformer_amd import dp, synthetic
    torch.set_num_threads(1)
    dp.init_from_env(backend="gloo")
    assert dp.world_size() == 4
    lens = [9, 7, 8, 6]
    build = lambda ang, seq: torch.stack([                                     # noqa: E731
        torch.cat([geometry.generate_coords(ang[b, :n], seq[b, :n]), torch.zeros((seq.shape[1] - n) * 14, 3)])
        for b, n in enumerate(lens)])
    batch = synthetic.make_batch(lens, seed=9, build_coords=build)
    am = synthetic.angle_means(batch["true_ang"])
    params = oenc.init_params(1, 32, 64, 16, am, seed=3)
    params["output_projection.weight"].normal_(0, 0.05)
    names = [k for k in params if not k.endswith(".pe")]

    def flat_grad(idx):
        leaf = {k: params[k].clone().requires_grad_() for k in names}
        if not idx:
            return torch.zeros(sum(v.numel() for v in leaf.values()))
        seq, crd = batch["seq"][idx], batch["true_crd"][idx]
        pred = oenc.encoder_forward({**leaf, "encoder.positional_enc.pe": params["encoder.positional_enc.pe"]}, seq, 4)
        olosses.compute_batch_drmsd(pred, crd, seq, do_backward=True)
        return torch.cat([leaf[k].grad.reshape(-1) for k in names])

    mine = [[0, 1], [2], [3], []][rank]
    g = flat_grad(mine)
    n = g.numel()
    model = types.SimpleNamespace(_flat=None, _flat_grad=g.clone(), grad_hook=None)
    model.flat_parameters = lambda: (model._flat, model._flat_grad)
    model.grad_slices = lambda: [(0, n // 3), (n // 3, n // 3), (2 * (n // 3), n - 2 * (n // 3))]
    dp.attach(model)
    if mine:                                     # "backward" reports its slices as they become final
        for off, cnt in model.grad_slices():
            model.grad_hook(off, cnt)
    dp.all_reduce_gradients(model, empty=not mine)
    full = flat_grad([0, 1, 2, 3])
    assert float((model._flat_grad - full).abs().max()) <= 1e-5 * float(full.abs().max())
    # every rank ends with the same bits
    gathered = [torch.zeros_like(full) for _ in range(4)]
    torch.distributed.all_gather(gathered, model._flat_grad)
    assert all(torch.equal(gathered[0], t) for t in gathered)
    assert sum(c for _, c in model.grad_slices()) == n
    if rank == 0:
        np.save(os.path.join(out_dir, "ok4.npy"), np.array([1]))
    dp.barrier()
    dp.shutdown()


def test_dp_world_size_4_with_an_empty_shard(tmp_path):
    port = _free_port()
    mp.spawn(_worker4, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    assert os.path.exists(tmp_path / "ok4.npy")
