"""Data-parallel end-to-end checks on ONE MI355X: two ranks share cuda:0 and exchange gradients and loss statistics
through gloo (RCCL refuses two ranks on one device; the 8-GPU RCCL run belongs to the driver).  The sharded,
hook-overlapped, SUM-reduced step must reproduce the single-process step on the whole batch:

  * for every training loss - `drmsd` (sum over proteins), `mse` and `combined` (means over the GLOBAL batch's selected
    angles: numerator and count are reduced before the gradient is formed) - with UNEQUAL shards (5 ragged proteins dealt
    in serpentine order of length: 3 + 2), identical losses on both ranks and identical updated parameters;
  * with an EMPTY shard (1 protein, 2 ranks): the idle rank joins every collective and ends with the same parameters;
  * `eval_epoch` sharded over the ranks reports the single-process metrics.
"""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

LENS = {"ragged": [40, 33, 48, 21, 37], "single": [29], "equal": [40, 33, 48, 21, 37, 48]}
KEYS = ("loss", "drmsd-full", "lndrmsd-full", "drmsd-bb", "lndrmsd-bb", "combined-full", "mse-full", "mse-bb", "mse-sc")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(dev, loss, case, conv=False):
    from protein_transformer_amd import synthetic
    from protein_transformer_amd.models.convolutional_encoder import ConvEncoderOnlyTransformer
    from protein_transformer_amd.models.encoder_only import EncoderOnlyTransformer
    from protein_transformer_amd.optim import FusedSGD
    from protein_transformer_amd.protein.Sequence import VOCAB
    from protein_transformer_amd.protein.Structure import nerf_forward
    build = lambda ang, seq: nerf_forward(ang.to(dev), seq.to(dev))[0]           # noqa: E731
    batch = synthetic.make_batch(LENS[case], L_pad=48, seed=9, build_coords=build, frac_missing=0.05)
    am = synthetic.angle_means(batch["true_ang"])
    torch.manual_seed(123)
    if conv:     # `-m "conv-enc|3,5|2,2"`: the Conv1d windows reach across a protein's end into the padding
        model = ConvEncoderOnlyTransformer(2, 4, 64, 128, 64, VOCAB, am, True, [3, 5], [2, 2], True, True, dropout=0.0)
    else:
        model = EncoderOnlyTransformer(2, 4, 64, 128, 64, VOCAB, am, True, dropout=0.0)
    with torch.no_grad():
        model.output_projection.weight.normal_(0, 0.05)
    model.set_dropout(0.0)
    model = model.to(dev).train()
    opt = FusedSGD(model, lr=1e-2, weight_decay=10e-3)
    args = types.SimpleNamespace(loss=loss, combined_drmsd_weight=0.5, backbone_loss=False, clip=1.0, lr_scheduling="plateau")
    return model, opt, args, tuple(batch[k] for k in ("seq", "true_ang", "true_crd")), LENS[case]


def _shard(batch, lens, world, rank, pad_to_global=False):
    """What dataset.ShardedBatchSampler + collate hand to a rank: its proteins, padded to its own longest one or (the
    sampler's default, `pad_to_global`) to the longest protein of the global batch."""
    from protein_transformer_amd import dp
    keep = dp.shard_indices(lens, world, rank)
    if not keep:
        return (torch.zeros(0, 0, dtype=torch.int64), torch.zeros(0, 0, 24), torch.zeros(0, 0, 3)), 0
    Lr = max(lens) if pad_to_global else max(lens[i] for i in keep)
    seq, ang, crd = batch
    return (seq[keep, :Lr].contiguous(), ang[keep, :Lr].contiguous(), crd[keep, :Lr * 14].contiguous()), sum(lens[i] for i in keep)


def _worker(rank, world, port, out_dir, loss, case, backend="gloo"):
    # gloo: both ranks on cuda:0 (RCCL refuses two ranks per device); nccl (= RCCL): one GPU per rank
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank) if backend == "nccl" else "0", PTAMD_DIST_BACKEND=backend,
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from protein_transformer_amd import dp
    from protein_transformer_amd.log import init_metrics
    from protein_transformer_amd.train import eval_epoch, train_step
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    dp.init_from_env()
    assert torch.distributed.get_backend() == backend
    conv = case.startswith("conv-")
    case = case[5:] if conv else case
    model, opt, args, batch, lens = _make(dev, loss, case, conv)
    dp.attach(model)
    (seq, ang, crd), n_res = _shard(batch, lens, world, rank, pad_to_global=conv)
    if case == "ragged":
        assert seq.shape[0] == (3 if rank == 0 else 2)
    if case == "single":
        assert seq.shape[0] == (1 if rank == 0 else 0)
    losses = train_step(model, opt, args, seq.to(dev), ang.to(dev), crd.to(dev), n_res=n_res)
    assert losses["n-residues"] == sum(lens)
    flat, _ = model.flat_parameters()
    np.save(os.path.join(out_dir, f"flat{rank}.npy"), flat.cpu().numpy())
    np.save(os.path.join(out_dir, f"loss{rank}.npy"), np.array([float(losses[k]) for k in KEYS]))
    # sharded evaluation of the same batch with the updated model
    metrics = eval_epoch(model, [(seq, ang, crd)], dev, args, init_metrics(args), mode="valid-70")
    m = metrics["valid-70"]
    np.save(os.path.join(out_dir, f"eval{rank}.npy"), np.array([m["epoch-drmsd-full"], m["epoch-lndrmsd-full"],
                                                                m["epoch-mse-full"], m["epoch-rmsd-full"]]))
    dp.barrier()
    dp.shutdown()


def test_two_rank_step_on_two_gpus_nccl(tmp_path):
    """The same check over RCCL, one GPU per rank - the asynchronous path gloo cannot exercise (per-layer all-reduce issued
    from the backward pass on RCCL's stream, `work.wait()` on the compute stream).  Needs two GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU node); one GPU: the gloo variants below")
    test_two_rank_step_equals_full_batch(tmp_path, "combined", "ragged", backend="nccl")


@pytest.mark.parametrize("loss,case", [("drmsd", "ragged"), ("combined", "ragged"), ("mse", "ragged"), ("lndrmsd", "equal"),
                                       ("combined", "single"), ("combined", "conv-ragged")])
def test_two_rank_step_equals_full_batch(tmp_path, loss, case, backend="gloo"):
    """`conv-ragged`: a conv-enc model (Conv1d windows cross the end of a protein) with the shards padded to the longest
    protein of the GLOBAL batch, as dataset.ShardedBatchSampler does: only then does a rank's longest protein see the same
    columns behind its end as in the single-process batch."""
    assert torch.cuda.is_available()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), loss, case, backend), nprocs=2, join=True)
    from protein_transformer_amd.log import init_metrics
    from protein_transformer_amd.train import eval_epoch, train_step
    dev = torch.device("cuda:0")
    conv = case.startswith("conv-")
    case = case[5:] if conv else case
    model, opt, args, batch, lens = _make(dev, loss, case, conv)
    start = model.flat_parameters()[0].cpu().numpy().copy()
    data = tuple(t.to(dev) for t in batch)
    losses = train_step(model, opt, args, *data)
    full = model.flat_parameters()[0].cpu().numpy()
    f0, f1 = np.load(tmp_path / "flat0.npy"), np.load(tmp_path / "flat1.npy")
    assert np.array_equal(f0, f1)                                   # ranks stay in lock step (also the idle one)
    upd, upd_dp = full - start, f0 - start
    assert np.linalg.norm(upd) > 0
    assert np.linalg.norm(upd_dp - upd) <= 1e-4 * np.linalg.norm(upd)
    l0, l1 = np.load(tmp_path / "loss0.npy"), np.load(tmp_path / "loss1.npy")
    assert np.array_equal(l0, l1)                                   # every rank reports the GLOBAL statistics
    want = np.array([float(losses[k]) for k in KEYS])
    assert l0 == pytest.approx(want, rel=1e-5, abs=1e-7)
    metrics = eval_epoch(model, [batch], dev, args, init_metrics(args), mode="valid-70")
    m = metrics["valid-70"]
    e0, e1 = np.load(tmp_path / "eval0.npy"), np.load(tmp_path / "eval1.npy")
    assert np.array_equal(e0, e1)
    assert e0 == pytest.approx(np.array([m["epoch-drmsd-full"], m["epoch-lndrmsd-full"], m["epoch-mse-full"],
                                         m["epoch-rmsd-full"]]), rel=2e-4)
