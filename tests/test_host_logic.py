"""CPU-only tests of the host-side mirror: vocabulary, CLI surface, Noam schedule, metrics, synthetic data."""
import numpy as np
import torch
from pytest import approx

from protein_transformer_amd import synthetic
from protein_transformer_amd.protein.Sequence import AA_MAP, VOCAB, ProteinVocabulary


def test_vocabulary():
    # /root/reference/protein_transformer/protein/Sequence.py:1-91
    assert len(VOCAB) == 22 and VOCAB.pad_id == 20 and VOCAB.sos_id == VOCAB.eos_id == 21
    assert VOCAB.str2ints("ACDEFGHIKLMNPQRSTVWY", add_sos_eos=False) == list(range(20))
    assert VOCAB.ints2str([0, 19, 20, 21]) == "AY?"
    assert VOCAB.int2chars(18) == "TRP" and AA_MAP["TRP"] == 18 and VOCAB["X"] == 21
    v = ProteinVocabulary(add_sos_eos=True)
    assert len(v) == 24 and v.sos_id == 22 and v.eos_id == 23
    assert v.str2ints("AC") == [22, 0, 1, 23]


def test_cli_defaults_match_reference():
    # SURVEY.md appendix G (defaults verified by running the reference's create_parser())
    from protein_transformer_amd.train import create_parser, parse_conv_kernel_info_from_model_name
    a = create_parser().parse_args([])
    assert (a.learning_rate, a.epochs, a.batch_size, a.early_stopping, a.n_warmup_steps, a.clip) == (1e-4, 10, 8, 20, 10000, 1)
    assert (a.loss, a.optimizer, a.lr_scheduling, a.patience, a.seed) == ("combined", "sgd", "plateau", 10, 11731)
    assert (a.model, a.d_model, a.d_inner_hid, a.n_head, a.n_layers, a.dropout) == ("enc-only", 512, 2048, 8, 6, 0.1)
    assert a.weight_decay is True and a.combined_drmsd_weight == 0.5 and a.bins == -1 and a.repeat_train == 1
    assert a.batching_order == "binned-random" and a.train_eval_downsample == 0.1 and a.checkpoint_time_interval == 0
    a = create_parser().parse_args("-m enc-only -dm 64 -nl 2 -b 4 -l drmsd --no_cuda --weight_decay False".split())
    assert a.d_model == 64 and a.n_layers == 2 and a.batch_size == 4 and a.no_cuda and a.weight_decay is False
    assert parse_conv_kernel_info_from_model_name("conv-enc|3,7,11|2,2,2") == ([3, 7, 11], [2.0, 2.0, 2.0])
    assert parse_conv_kernel_info_from_model_name("conv-enc") == ([], [])


def test_noam_schedule():
    # models/transformer/Optimizer.py:4-62
    from protein_transformer_amd.optim import ScheduledOptim

    class Dummy:
        param_groups = [{"lr": 0.0}]
        steps = 0

        def step(self):
            self.steps += 1
    opt = ScheduledOptim(Dummy(), 512, 4000)
    lrs = []
    for _ in range(3):
        opt.step()
        lrs.append(opt.param_groups[0]["lr"])
    assert lrs[0] == approx(512 ** -0.5 * 4000 ** -1.5) and lrs[2] == approx(3 * lrs[0])
    opt.n_current_steps = 10 ** 6
    opt.step()
    assert opt.param_groups[0]["lr"] == approx(512 ** -0.5 * (10 ** 6 + 1) ** -0.5)
    sd = opt.state_dict if False else None  # state_dict needs a real optimizer; covered on the GPU


def test_metrics_and_speed():
    import types
    from protein_transformer_amd import log
    args = types.SimpleNamespace(lr_scheduling="plateau", loss="drmsd", es_mode="train", es_metric="drmsd",
                                 early_stopping=2, early_stopping_threshold=0.001)
    m = log.init_metrics(args)
    assert m["history-lr"] == [0] and "valid-70" in m
    m = log.reset_metrics_for_epoch(m, "train")
    seq = torch.full((2, 10), 20)
    seq[:, :6] = 1
    losses = {"loss": 1.0, "drmsd-full": 2.0, "lndrmsd-full": 0.1, "drmsd-bb": 1.0, "lndrmsd-bb": 0.2, "combined-full": 3.0,
              "mse-full": torch.tensor(0.04), "mse-bb": 0.1, "mse-sc": 0.2, "rmsd-full": None}
    m = log.update_metrics(m, losses, "train", seq, tracking_loss=1.0)
    assert m["train"]["speed"] > 0 and m["train"]["batch-drmsd-full"] == 2.0 and m["n_batches"] == 1
    m = log.update_metrics_end_of_epoch(m, "train")
    assert m["train"]["epoch-history-drmsd"] == [2.0]
    m = log.update_loss_trackers(args, 0, m)
    assert m["best_valid_loss_so_far"] == 2.0 and m["epoch_last_improved"] == 0
    assert log.prepare_log_header(args).split(",")[:4] == ["drmsd", "ln_drmsd", "rmse", "rmsd"]


def test_synthetic_batch():
    from oracle import geometry
    lens = [6, 9]
    build = lambda ang, seq: torch.stack([                                     # noqa: E731
        torch.cat([geometry.generate_coords(ang[b, :n], seq[b, :n]), torch.zeros((seq.shape[1] - n) * 14, 3)])
        for b, n in enumerate(lens)])
    b = synthetic.make_batch(lens, seed=3, build_coords=build)
    assert b["seq"].shape == (2, 9) and int((b["seq"][0] == 20).sum()) == 3
    own = synthetic.slot_mask(b["seq"])
    crd = b["true_crd"]
    assert torch.isnan(crd[0, :6 * 14][~own[0, :6 * 14]]).all()               # unused slots are NaN
    assert not torch.isnan(crd[0, :6 * 14][own[0, :6 * 14]]).any()
    assert (crd[0, 6 * 14:] == 0).all()                                       # batch padding is zeros
    am = synthetic.angle_means(b["true_ang"])
    assert am.shape == (24,) and np.all(np.abs(am) <= 1)
    b2 = synthetic.make_batch(lens, seed=3)
    assert torch.equal(b["seq"], b2["seq"])                                   # deterministic
