"""Data-parallel end-to-end check on ONE MI355X: two ranks share cuda:0 and exchange gradients through gloo
(RCCL refuses two ranks on one device; the 8-GPU RCCL run belongs to the driver).  The sharded, hook-overlapped,
SUM-reduced step must reproduce the single-process step on the whole batch: identical losses per shard and
identical updated parameters."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(dev):
    from protein_transformer_amd import synthetic
    from protein_transformer_amd.models.encoder_only import EncoderOnlyTransformer
    from protein_transformer_amd.optim import FusedSGD
    from protein_transformer_amd.protein.Sequence import VOCAB
    from protein_transformer_amd.protein.Structure import nerf_forward
    build = lambda ang, seq: nerf_forward(ang.to(dev), seq.to(dev))[0]           # noqa: E731
    batch = synthetic.make_batch([40, 33, 48, 21, 37, 48], L_pad=48, seed=9, build_coords=build)
    am = synthetic.angle_means(batch["true_ang"])
    torch.manual_seed(123)
    model = EncoderOnlyTransformer(2, 4, 64, 128, 64, VOCAB, am, True, dropout=0.0)
    with torch.no_grad():
        model.output_projection.weight.normal_(0, 0.05)
    model.set_dropout(0.0)
    model = model.to(dev).train()
    opt = FusedSGD(model, lr=1e-2, weight_decay=10e-3)
    args = types.SimpleNamespace(loss="drmsd", combined_drmsd_weight=0.5, backbone_loss=False, clip=1.0)
    return model, opt, args, tuple(batch[k].to(dev) for k in ("seq", "true_ang", "true_crd"))


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", PTAMD_DIST_BACKEND="gloo")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from protein_transformer_amd import dp
    from protein_transformer_amd.train import train_step
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dp.init_from_env()
    model, opt, args, batch = _make(dev)
    dp.attach(model)
    shard = dp.shard_batch(*batch)
    assert shard[0].shape[0] == 3
    losses = train_step(model, opt, args, *shard)
    flat, _ = model.flat_parameters()
    np.save(os.path.join(out_dir, f"flat{rank}.npy"), flat.cpu().numpy())
    np.save(os.path.join(out_dir, f"loss{rank}.npy"), np.array([losses["drmsd-full"], losses["lndrmsd-full"]]))
    dp.barrier()
    dp.shutdown()


def test_two_rank_step_equals_full_batch(tmp_path):
    assert torch.cuda.is_available()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    from protein_transformer_amd.train import train_step
    dev = torch.device("cuda:0")
    model, opt, args, batch = _make(dev)
    losses = train_step(model, opt, args, *batch)
    full = model.flat_parameters()[0].cpu().numpy()
    f0, f1 = np.load(tmp_path / "flat0.npy"), np.load(tmp_path / "flat1.npy")
    assert np.array_equal(f0, f1)                                   # ranks stay in lock step
    model0, _, _, _ = _make(dev)
    start = model0.flat_parameters()[0].cpu().numpy()
    upd, upd_dp = full - start, f0 - start
    assert np.linalg.norm(upd_dp - upd) <= 1e-4 * np.linalg.norm(upd)
    l0, l1 = np.load(tmp_path / "loss0.npy"), np.load(tmp_path / "loss1.npy")
    assert (l0[0] + l1[0]) / 2 == pytest.approx(float(losses["drmsd-full"]), rel=1e-5)   # equal shard sizes
