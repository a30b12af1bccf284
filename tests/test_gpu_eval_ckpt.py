"""MI355X tests of SURVEY.md section 8(f) rows 1 and 3 and of the boundary's re-entrancy (8b):

  * `rmsd` (losses.py:281-286): the batched device Kabsch kernel against `oracle.losses.kabsch_rmsd` (the textbook
    superposition ProDy implements; parity unpinned, ProDy is not installable) on the structures of golden G4, on
    rotated / translated / mirrored copies with known answers, and through `drmsd_work(return_rmsd=True)`;
  * `eval_epoch` (train.py:114-135): epoch means of drmsd / lndrmsd / mse / rmsd against the CPU oracle per protein;
  * checkpoint -> resume (train.py:189-271): model, Adam moments and step, Noam tuple, START_EPOCH, elapsed time; the
    reference's best-else-latest policy and signature; the CSV `.train` log (log.py:115-130,488-495);
  * two models with different arithmetics interleaved on two HIP streams give the results they give alone.
"""
import csv
import os
import types

import numpy as np
import pytest
import torch
from pytest import approx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def T(x):
    return torch.tensor(np.asarray(x))


def _small_batch(lens, L_pad, seed, frac_missing=0.05):
    from oracle import geometry
    from protein_transformer_amd import synthetic
    build_cpu = lambda ang, seq: torch.stack([                                  # noqa: E731
        torch.cat([geometry.generate_coords(ang[b, :n], seq[b, :n]), torch.zeros((seq.shape[1] - n) * 14, 3)])
        for b, n in enumerate(lens)])
    return synthetic.make_batch(lens, L_pad=L_pad, seed=seed, build_coords=build_cpu, frac_missing=frac_missing)


# ------------------------------------------------------------------------------------------------ Kabsch RMSD
def test_kabsch_rmsd_vs_oracle_on_g4(golden, dev):
    from oracle import geometry, losses as olosses
    from protein_transformer_amd.eval_metrics import kabsch_rmsd_batch, rmsd
    from protein_transformer_amd.losses import drmsd_work
    g = golden("g4_drmsd_work")
    seq, ang, true = T(g["seq"]), T(g["pred_ang"]), T(g["true_crd"])
    B, L = seq.shape
    pred = torch.zeros(B, L * 14, 3)
    want = []
    for b in range(B):
        n = int((seq[b] != 20).sum())
        crd = geometry.generate_coords(ang[b, :n], seq[b, :n])
        pred[b, :n * 14] = crd
        t = true[b, :n * 14]
        ok = ~torch.isnan(t).any(1)
        want.append(olosses.kabsch_rmsd(crd[ok].numpy(), t[ok].numpy()))
    got = kabsch_rmsd_batch(pred.to(dev), true.to(dev), seq.to(dev)).cpu().numpy()
    assert got == approx(np.array(want), rel=2e-6, abs=1e-6)
    # the reference's call style: drmsd_work(..., return_rmsd=True) appends the value (losses.py:94-96)
    r = drmsd_work(g["pred_ang"][1], g["true_crd"][1], g["seq"][1], return_rmsd=True)
    assert len(r) == 6 and r[5] == approx(want[1], rel=1e-4)
    n = int((seq[0] != 20).sum()) * 14
    t = true[0, :n]
    ok = ~torch.isnan(t).any(1)
    assert rmsd(pred[0, :n][ok], t[ok]) == approx(want[0], rel=2e-6)


def test_kabsch_known_answers(dev):
    """Rigid motions give 0, a mirror image does not, isotropic noise gives about its sigma; absent atoms and padded
    residues are ignored; a protein without atoms reports NaN."""
    from oracle import losses as olosses
    from protein_transformer_amd.eval_metrics import kabsch_rmsd_batch
    rng = np.random.default_rng(5)
    L = 40
    a = rng.normal(0, 12, (L * 14, 3))
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    moved = a @ q.T + np.array([30.0, -7.0, 110.0])
    mirrored = a * np.array([1.0, 1.0, -1.0])
    noisy = moved + rng.normal(0, 0.5, a.shape)
    holes = moved.copy()
    holes[rng.random(L * 14) < 0.3] = np.nan
    short = moved.copy()
    pred = np.stack([a] * 6)
    true = np.stack([moved, mirrored, noisy, holes, short, np.full_like(a, np.nan)])
    seq = torch.zeros(6, L, dtype=torch.int64)
    seq[4, 25:] = 20                                             # only the first 25 residues exist
    true[4, 25 * 14:] = 0                                        # collate pads with zeros
    pred[4, 25 * 14:] = 0
    got = kabsch_rmsd_batch(T(pred).float().to(dev), T(true).float().to(dev), seq.to(dev)).cpu().numpy()
    assert got[0] < 2e-5 and got[3] < 2e-5 and got[4] < 2e-5      # fp32 coordinates of size ~100
    assert got[1] == approx(olosses.kabsch_rmsd(a, mirrored), rel=1e-5) and got[1] > 1.0
    assert got[2] == approx(olosses.kabsch_rmsd(a, noisy), rel=1e-4)
    assert 0.7 < got[2] < 1.0                                      # sqrt(3) * 0.5 = 0.87 minus the fitted part
    assert np.isnan(got[5])


# ------------------------------------------------------------------------------------------------ eval_epoch
def _tiny_model(dev, am, optimizer="sgd", seed=3, dropout=0.0, nl=2):
    from protein_transformer_amd.models.encoder_only import EncoderOnlyTransformer
    from protein_transformer_amd.protein.Sequence import VOCAB
    torch.manual_seed(seed)
    m = EncoderOnlyTransformer(nl, 4, 64, 128, 64, VOCAB, am, True, dropout=dropout)
    with torch.no_grad():
        m.output_projection.weight.normal_(0, 0.05)
    if dropout == 0.0:
        m.set_dropout(0.0)
    return m.to(dev)


def test_eval_epoch_vs_oracle(dev):
    from oracle import encoder as oenc, losses as olosses
    from protein_transformer_amd import synthetic
    from protein_transformer_amd.log import init_metrics
    from protein_transformer_amd.train import eval_epoch
    lens = [30, 17, 24, 9, 28, 21]
    batch = _small_batch(lens, 30, seed=8)
    am = synthetic.angle_means(batch["true_ang"])
    model = _tiny_model(dev, am)
    args = types.SimpleNamespace(loss="drmsd", combined_drmsd_weight=0.5, backbone_loss=False, lr_scheduling="plateau")
    loader = [tuple(batch[k][i:i + 2] for k in ("seq", "true_ang", "true_crd")) for i in (0, 2, 4)]   # 3 batches of 2
    metrics = eval_epoch(model, loader, dev, args, init_metrics(args), mode="valid-70")
    assert not model.training
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    want = {"drmsd": [], "ln": [], "rmsd": [], "mse": []}
    with torch.no_grad():
        for seq, ang, crd in loader:
            pred = oenc.encoder_forward(params, seq, 4)
            rad = olosses.inverse_trig_transform(pred)
            per = [olosses.drmsd_work(rad[b].numpy(), crd[b].numpy(), seq[b].numpy(), return_rmsd=True, do_backward=False)
                   for b in range(seq.shape[0])]
            want["drmsd"].append(np.mean([p[1] for p in per]))
            want["ln"].append(np.mean([p[2] for p in per]))
            want["rmsd"].append(np.mean([p[5] for p in per]))
            want["mse"].append(float(olosses.mse_over_angles(pred, ang)))
    m = metrics["valid-70"]
    assert m["epoch-drmsd-full"] == approx(np.mean(want["drmsd"]), rel=1e-4)
    assert m["epoch-lndrmsd-full"] == approx(np.mean(want["ln"]), rel=1e-4)
    assert m["epoch-rmsd-full"] == approx(np.mean(want["rmsd"]), rel=1e-4)
    assert m["epoch-mse-full"] == approx(np.mean(want["mse"]), rel=1e-5)
    assert m["epoch-history-drmsd"] == [m["epoch-drmsd-full"]]


# ------------------------------------------------------------------------------------------------ checkpoint / resume
def _ckpt_args(tmp_path, **kw):
    d = dict(loss="drmsd", combined_drmsd_weight=0.5, backbone_loss=False, clip=1.0, lr_scheduling="noam",
             checkpoint_time_interval=0, chkpt_path=str(tmp_path / "model"), restart=False, restart_opt=False,
             load_chkpt=None, d_model=64, n_warmup_steps=10)
    d.update(kw)
    return types.SimpleNamespace(**d)


@pytest.mark.parametrize("opt_name", ["adam-noam", "sgd-plateau"])
def test_checkpoint_resume_roundtrip(dev, tmp_path, opt_name):
    """Train 3 steps, checkpoint, train 2 more (A).  A fresh process state resumed from the checkpoint and trained for
    the same 2 steps (B) must land on the same parameters: model, optimizer moments + step count, the Noam tuple and
    START_EPOCH / elapsed time all survive the round trip (the Adam moments used to be zeroed)."""
    from protein_transformer_amd import synthetic, train as TR
    from protein_transformer_amd.optim import FusedAdam, FusedSGD, ScheduledOptim
    batch = _small_batch([30, 17, 24, 9], 30, seed=12)
    am = synthetic.angle_means(batch["true_ang"])
    data = tuple(batch[k].to(dev) for k in ("seq", "true_ang", "true_crd"))
    noam = opt_name == "adam-noam"
    args = _ckpt_args(tmp_path, lr_scheduling="noam" if noam else "plateau")

    def make():
        model = _tiny_model(dev, am, seed=4).train()
        if noam:
            opt = ScheduledOptim(FusedAdam(model, betas=(0.9, 0.98), eps=1e-9, lr=1e-3, weight_decay=10e-3), 64, 10)
            sched = None
        else:
            opt = FusedSGD(model, lr=1e-2, weight_decay=10e-3)
            sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=1, threshold=0.001)
        return model, opt, sched

    model, opt, sched = make()
    for _ in range(3):
        TR.train_step(model, opt, args, *data)
    if sched:
        sched.step(5.0); sched.step(6.0); sched.step(7.0)            # the plateau scheduler has lowered the rate once
    metrics = {"loss_to_compare": 1.5, "losses_to_compare": [1.5], "last_chkpt_time": 0.0, "marker": "kept"}
    TR.START_TIME -= 100.0                                           # pretend the run is 100 s old
    assert TR.checkpoint_model(args, opt, model, metrics, 4, sched) is True
    assert os.path.exists(args.chkpt_path + "_best.chkpt") and not os.path.exists(args.chkpt_path + "_latest.chkpt")
    for _ in range(2):
        TR.train_step(model, opt, args, *data)
    want = model.flat_parameters()[0].cpu().numpy().copy()
    lr_a = opt.param_groups[0]["lr"]

    model2, opt2, sched2 = make()
    TR.START_EPOCH, TR.START_TIME = 0, 1e9
    t_before = TR.START_TIME
    model2, opt2, sched2, resumed, metrics2 = TR.load_model(model2, opt2, sched2, args)
    assert resumed and metrics2["marker"] == "kept" and TR.START_EPOCH == 5
    assert t_before - TR.START_TIME >= 100.0                         # elapsed time stays cumulative
    if noam:
        assert opt2.n_current_steps == 3 and opt2._optimizer._t == 3
        assert opt2._optimizer._m.device.type == "cuda" and float(opt2._optimizer._m.abs().sum()) > 0
    else:
        assert opt2.param_groups[0]["lr"] == approx(1e-3)            # the reduced rate came back
        assert sched2.state_dict()["num_bad_epochs"] == sched.state_dict()["num_bad_epochs"]
    for _ in range(2):
        TR.train_step(model2.train(), opt2, args, *data)
    got = model2.flat_parameters()[0].cpu().numpy()
    assert opt2.param_groups[0]["lr"] == approx(lr_a)
    assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max()
    # --restart ignores the file; --restart_opt keeps the model but not the optimizer
    m3, o3, s3 = make()
    _, _, _, resumed3, metrics3 = TR.load_model(m3, o3, s3, _ckpt_args(tmp_path, restart=True, lr_scheduling=args.lr_scheduling))
    assert not resumed3 and "train" in metrics3
    m4, o4, s4 = make()
    TR.load_model(m4, o4, s4, _ckpt_args(tmp_path, restart_opt=True, lr_scheduling=args.lr_scheduling))
    if noam:
        assert o4.n_current_steps == 0 and o4._optimizer._t == 0
    TR.START_EPOCH = 0


def test_checkpoint_policy_and_signature(dev, tmp_path):
    """train.py:189-230: `checkpoint_model(args, optimizer, model, metrics, epoch_i, scheduler)`; best when the loss beats
    all earlier ones, else latest when the time interval has passed, else nothing."""
    import inspect
    import time
    from protein_transformer_amd import train as TR
    from protein_transformer_amd.optim import FusedSGD
    assert list(inspect.signature(TR.checkpoint_model).parameters) == ["args", "optimizer", "model", "metrics", "epoch_i", "scheduler"]
    model = _tiny_model(dev, np.zeros(24) + 0.2, nl=1)
    opt = FusedSGD(model, lr=1e-2)
    args = _ckpt_args(tmp_path, checkpoint_time_interval=1.0)
    best, latest = args.chkpt_path + "_best.chkpt", args.chkpt_path + "_latest.chkpt"
    now = time.time()
    m = {"loss_to_compare": 2.0, "losses_to_compare": [2.0], "last_chkpt_time": now}
    assert TR.checkpoint_model(args, opt, model, m, 0, None) and os.path.exists(best)
    assert torch.load(best, weights_only=False)["epoch"] == 0
    m = {"loss_to_compare": 3.0, "losses_to_compare": [2.0, 3.0], "last_chkpt_time": now}
    assert TR.checkpoint_model(args, opt, model, m, 1, None) is False and not os.path.exists(latest)   # worse, too early
    m = {"loss_to_compare": 3.0, "losses_to_compare": [2.0, 3.0], "last_chkpt_time": now - 2 * 3600}
    assert TR.checkpoint_model(args, opt, model, m, 2, None) and os.path.exists(latest)                # worse, interval passed
    assert m["last_chkpt_time"] >= now and torch.load(best, weights_only=False)["epoch"] == 0           # best untouched
    m = {"loss_to_compare": 1.0, "losses_to_compare": [2.0, 3.0, 1.0], "last_chkpt_time": now - 2 * 3600}
    assert TR.checkpoint_model(args, opt, model, m, 3, None)
    ck = torch.load(best, weights_only=False)
    assert ck["epoch"] == 3 and ck["loss"] == 1.0 and set(ck) == {"model_state_dict", "settings", "epoch", "optimizer_state_dict",
                                                                  "scheduler_state_dict", "loss", "metrics", "elapsed_time"}
    assert set(ck["model_state_dict"]) == set(model.state_dict())


def test_train_cli_writes_log_and_resumes(dev, tmp_path, monkeypatch):
    """`python -m protein_transformer_amd.train --synthetic ...` end to end, twice: the second invocation resumes from
    `<name>_best.chkpt`, appends to the `.train` CSV (header once, cumulative time column) and starts at the next epoch."""
    import sys
    from protein_transformer_amd import train as TR
    common = ["train", "--synthetic", "4,24,2", "--name", "t1", "-dm", "64", "-nl", "1", "-nh", "4", "-dih", "128",
              "-l", "drmsd", "-b", "4", "--max_seq_len", "24", "--train_only", "--log_dir", str(tmp_path / "logs"),
              "--chkpt_dir", str(tmp_path / "ck"), "-opt", "adam"]
    monkeypatch.setattr(TR, "START_EPOCH", 0)
    monkeypatch.setattr(sys, "argv", common + ["-e", "2"])
    TR.main()
    log = tmp_path / "logs" / "t1.train"
    rows = list(csv.reader(open(log)))
    assert rows[0] == "drmsd,ln_drmsd,rmse,rmsd,lr,mode,granularity,time,speed".split(",")
    # rows carry ten values - the `combined` column is always written (log.py:128-130), whatever the header says
    DR, LN, RMSE, RMSD, COMB, LR, MODE, GRAN, TIME, SPEED = range(10)
    assert all(len(r) == 10 for r in rows[1:])
    n_first = len(rows)
    epochs = [r for r in rows[1:] if r[GRAN] == "epoch"]
    assert len(epochs) == 2 and all(r[MODE] == "train" for r in epochs)
    batches = [r for r in rows[1:] if r[GRAN] == "batch"]
    assert len(batches) == 4 and all(float(r[SPEED]) > 0 and float(r[DR]) > 0 for r in batches)   # residues / s
    assert os.path.exists(tmp_path / "ck" / "t1_best.chkpt")
    t_last = float(rows[-1][TIME])
    monkeypatch.setattr(sys, "argv", common + ["-e", "3"])
    TR.main()
    rows2 = list(csv.reader(open(log)))
    assert rows2[:n_first] == rows and sum(r[0] == "drmsd" for r in rows2) == 1  # appended, one header
    new_epochs = [r for r in rows2[n_first:] if r[GRAN] == "epoch"]
    assert 1 <= len(new_epochs) <= 2                                             # resumed behind the best epoch, not at 0
    assert float(rows2[n_first][TIME]) >= t_last * 0.3                           # the time column did not restart at zero
    monkeypatch.setattr(TR, "START_EPOCH", 0)


def test_adbs_probe_finds_a_batch_size_and_stops_at_the_memory_it_is_given(dev, tmp_path, monkeypatch):
    """`-adbs / --automatically_determine_batch_size` (train.py:532-551,586-587; scripts/determine_largest_batchsize.py): the
    probe doubles the batch size on largest-bin batches, measures the allocator peak of two training steps per candidate and
    stops where the next candidate would not fit.  (i) plenty of memory: the budget that covers the whole training set, times
    0.8; (ii) a pretended free memory just above the first probes' peak: the doubling stops early; (iii) the run's RNG
    streams are where they were; (iv) the flag is honoured by the CLI (round 5 parsed and ignored it)."""
    import sys
    from math import ceil
    from protein_transformer_amd import train as TR
    from protein_transformer_amd.synthetic_data import make_synthetic_dataset
    parser = TR.create_parser()
    args = parser.parse_args(["--synthetic", "8,48,4", "-dm", "64", "-nl", "1", "-nh", "4", "-dih", "128", "-l", "drmsd",
                              "-b", "2", "--max_seq_len", "48"])
    args.add_sos_eos, args.bins = False, "auto"
    data = make_synthetic_dataset(args.synthetic, args.seed, dev)
    am = data["settings"]["angle_means"]
    np.random.seed(5)
    torch.manual_seed(5)
    st_np, st_t = np.random.get_state()[1].copy(), torch.get_rng_state().clone()
    b_cap = ceil(32 * 48 / 48)
    got = TR.determine_largest_batch_size(args, data, dev, am)
    assert got == ceil(0.8 * b_cap)
    assert (np.random.get_state()[1] == st_np).all() and torch.equal(torch.get_rng_state(), st_t)
    # pretend the device has hardly any memory left beyond what two tiny probes need: the fit must stop the doubling
    seen = []
    real = torch.cuda.max_memory_allocated

    def peak(device=None):
        v = real(device)
        seen.append(v)
        return v
    monkeypatch.setattr(torch.cuda, "max_memory_allocated", peak)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda device=None: (int(0.02 * max(seen)) if seen else 1 << 40, 1 << 40))
    small = TR.determine_largest_batch_size(args, data, dev, am)
    assert 1 <= small < got
    monkeypatch.undo()
    # ... and through the command line
    monkeypatch.setattr(TR, "START_EPOCH", 0)
    monkeypatch.setattr(sys, "argv", ["train", "--synthetic", "8,48,2", "--name", "adbs", "-dm", "64", "-nl", "1", "-nh", "4",
                                      "-dih", "128", "-l", "drmsd", "-b", "1", "--max_seq_len", "48", "--train_only", "-e", "1",
                                      "-adbs", "True", "--log_dir", str(tmp_path / "logs"), "--chkpt_dir", str(tmp_path / "ck")])
    calls = []
    orig = TR.determine_largest_batch_size
    monkeypatch.setattr(TR, "determine_largest_batch_size", lambda *a, **k: calls.append(orig(*a, **k)) or calls[-1])
    TR.main()
    assert calls and calls[0] == ceil(0.8 * ceil(16 * 48 / 48))
    monkeypatch.setattr(TR, "START_EPOCH", 0)


# ------------------------------------------------------------------------------------------------ no global state
def test_two_models_two_streams_two_arithmetics(dev):
    """SURVEY.md section 8(b) 'no hidden global state, re-entrant': a model in the exact-f32 arithmetic and one in the
    default arithmetic, each on its own HIP stream, steps interleaved, produce what each produces alone."""
    from protein_transformer_amd import kernels as K_
    from protein_transformer_amd import synthetic
    from protein_transformer_amd.optim import FusedSGD
    from protein_transformer_amd.train import train_step
    batch = _small_batch([48, 31, 17, 40], 48, seed=5)
    am = synthetic.angle_means(batch["true_ang"])
    data = tuple(batch[k].to(dev) for k in ("seq", "true_ang", "true_crd"))
    args = types.SimpleNamespace(loss="combined", combined_drmsd_weight=0.5, backbone_loss=False, clip=1.0)

    def make(mode):
        m = _tiny_model(dev, am, seed=9, dropout=0.1).train()
        m.gemm_mode = mode
        return m, FusedSGD(m, lr=1e-2, weight_decay=10e-3)

    def alone(mode):
        m, o = make(mode)
        out = [float(train_step(m, o, args, *data)["loss"]) for _ in range(3)]
        return m.flat_parameters()[0].clone(), out

    ref = {mode: alone(mode) for mode in (K_.GEMM_F32, K_.GEMM_AUTO)}
    assert not torch.equal(ref[K_.GEMM_F32][0], ref[K_.GEMM_AUTO][0])          # the arithmetics do differ in the last bits
    torch.cuda.synchronize()
    streams = {K_.GEMM_F32: torch.cuda.Stream(), K_.GEMM_AUTO: torch.cuda.Stream()}
    models = {}
    for mode, st in streams.items():
        with torch.cuda.stream(st):
            models[mode] = make(mode)
    got = {mode: [] for mode in streams}
    for _ in range(3):
        for mode, st in streams.items():                                      # interleaved, concurrently in flight
            with torch.cuda.stream(st):
                m, o = models[mode]
                got[mode].append(float(train_step(m, o, args, *data)["loss"]))
    torch.cuda.synchronize()
    for mode in streams:
        assert got[mode] == ref[mode][1], mode
        assert torch.equal(models[mode][0].flat_parameters()[0], ref[mode][0]), mode
    assert K_.get_gemm_mode() == K_.GEMM_AUTO                                  # nobody touched the host default


def test_device_prefetcher(dev):
    """dataset.DevicePrefetcher: same batches, same order, on the device, with the host-side residue count; the next
    batch's copy is issued on a side stream before the current one is handed out."""
    from protein_transformer_amd.dataset import DevicePrefetcher
    g = torch.Generator().manual_seed(0)
    batches = []
    for n in (5, 3, 7, 1):
        seq = torch.randint(0, 21, (n, 12), generator=g)
        batches.append((seq.pin_memory(), torch.randn(n, 12, 24, generator=g).pin_memory(), torch.randn(n, 168, 3, generator=g).pin_memory()))
    got = list(DevicePrefetcher(batches, dev))
    assert len(got) == 4 and len(DevicePrefetcher(batches, dev)) == 4
    for (seq, ang, crd), (s, a, c, n_res) in zip(batches, got):
        assert s.is_cuda and torch.equal(s.cpu(), seq) and torch.equal(a.cpu(), ang) and torch.equal(c.cpu(), crd)
        assert n_res == int((seq != 20).sum())
    assert list(DevicePrefetcher([], dev)) == []
    # packed batches (dataset.pack_batch: what the collate function returns): ONE copy per batch, the same tensors on the
    # device - views of one buffer -, pinned by the prefetcher when the loader did not; slices of a packed batch too
    from protein_transformer_amd.dataset import pack_batch, packed_base
    for pin in (False, True):
        packed = [pack_batch(b, pin=pin) for b in batches] + [tuple(t[:2] for t in pack_batch(batches[2], pin=pin))]
        assert all(packed_base(b) is not None for b in packed) and packed_base(batches[0]) is None
        got = list(DevicePrefetcher(packed, dev))
        for (seq, ang, crd), (s, a, c, n_res) in zip(batches + [tuple(t[:2] for t in batches[2])], got):
            assert s.is_cuda and s.dtype == torch.int64 and torch.equal(s.cpu(), seq) and torch.equal(a.cpu(), ang) and torch.equal(c.cpu(), crd)
            assert s.untyped_storage().data_ptr() == a.untyped_storage().data_ptr() == c.untyped_storage().data_ptr()
            assert n_res == int((seq != 20).sum())
