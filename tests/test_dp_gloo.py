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
    model.grad_hook(0, n // 2)
    model.grad_hook(n // 2, n - n // 2)
    dp.all_reduce_gradients(model)
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
    dp.barrier()
    dp.shutdown()


def test_shard_bounds():
    from protein_transformer_amd.dp import shard_bounds
    assert [shard_bounds(32, 8, r) for r in range(8)] == [(4 * r, 4 * r + 4) for r in range(8)]
    assert [shard_bounds(5, 2, r) for r in range(2)] == [(0, 3), (3, 5)]
    assert [shard_bounds(3, 4, r) for r in range(4)] == [(0, 1), (1, 2), (2, 3), (3, 3)]


def test_dp_sum_allreduce_equals_full_batch(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got, full = np.load(tmp_path / "dp.npy"), np.load(tmp_path / "full.npy")
    assert np.abs(got - full).max() <= 1e-5 * np.abs(full).max()
