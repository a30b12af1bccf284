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
    # the granularity column: "batch" / "epoch" by default, upstream's literal "epoch" on every row with --reference-csv
    import csv
    import io
    for ref_csv, want in ((False, ["batch", "epoch"]), (True, ["epoch", "epoch"])):
        buf = io.StringIO()
        log.REFERENCE_CSV = ref_csv
        try:
            w = csv.writer(buf)
            log.log_batch(w, m, 0.0, mode="train", end_of_epoch=False, t=1.0)
            log.log_batch(w, m, 0.0, mode="train", end_of_epoch=True, t=2.0)
        finally:
            log.REFERENCE_CSV = False
        rows = list(csv.reader(io.StringIO(buf.getvalue())))
        assert [r[7] for r in rows] == want and all(len(r) == 10 and r[6] == "train" for r in rows)


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


# ------------------------------------------------------------------------------------------------ dropout generator
def _mix32(x):
    """numpy restatement of pt_mix32 (csrc/common.h)."""
    x = np.asarray(x).astype(np.uint32)
    x = (~x + (x << np.uint32(15))).astype(np.uint32)
    x ^= x >> np.uint32(12)
    x = (x + (x << np.uint32(2))).astype(np.uint32)
    x ^= x >> np.uint32(4)
    x = (x + (x << np.uint32(3)) + (x << np.uint32(11))).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def rand4_restated(seed, idx, stream):
    """numpy restatement of pt_rand4 (csrc/common.h): four uint32 arrays."""
    idx = np.asarray(idx, dtype=np.uint64)
    h = _mix32((idx & np.uint64(0xffffffff)).astype(np.uint32) ^ np.uint32(seed & 0xffffffff))
    h = _mix32(h ^ (idx >> np.uint64(32)).astype(np.uint32) ^ np.uint32((seed >> 32) & 0xffffffff)
               ^ np.uint32((stream * 0x9E3779B9) & 0xffffffff))
    return [_mix32(h ^ np.uint32(c)) for c in (0x68E31DA4, 0xB5297A4D, 0x1B56C4E9, 0x7F4A7C15)]


def dropout_mask_restated(rows, cols, p, seed, stream):
    """keep mask of the GEMM epilogue / ptamd_dropout_bwd (csrc/common.h: drop_call_index, drop_field): row
    32 I + 8 g + 4 h + e uses the 16-bit field (g & 1) * 4 + e of the call (I, h, g >> 1) of its column."""
    thr16 = np.uint32(np.uint32(min(p * 2.0 ** 32, 2.0 ** 32 - 1)) >> np.uint32(16))
    row = np.arange(rows, dtype=np.int64)[:, None]
    col = np.arange(cols, dtype=np.int64)[None, :]
    call_row = ((row >> 5) << 2) | (((row >> 2) & 1) << 1) | ((row >> 4) & 1)
    idx = (call_row * cols + col).astype(np.uint64)
    w = np.stack(rand4_restated(seed, idx, stream))                    # [4, rows, cols]
    f = (((row >> 3) & 1) * 4 + (row & 3)) + 0 * col                   # [rows, cols]
    word = np.take_along_axis(w, (f >> 1)[None].astype(np.int64), axis=0)[0]
    val = np.where(f & 1, word >> np.uint32(16), word & np.uint32(0xffff))
    return val >= thr16


def test_dropout_generator_statistics():
    """The multiply-free counter hash must give masks that look like independent Bernoulli draws."""
    rows, cols, p = 2048, 512, 0.1
    mask = dropout_mask_restated(rows, cols, p, 1234567890123, 7)
    n = mask.size
    assert abs(mask.mean() - (1 - p)) < 4 * np.sqrt(p * (1 - p) / n)
    m = mask.astype(np.float64) - mask.mean()

    def corr(a, b):
        return (a * b).mean() / (a.std() * b.std())
    for a, b in ((m[:, :-1], m[:, 1:]), (m[:-1], m[1:]), (m[:-4], m[4:]), (m[:-8], m[8:]), (m[:-1, :-1], m[1:, 1:])):
        assert abs(corr(a, b)) < 4 / np.sqrt(n)                      # neighbours, other fields of a call, diagonals
    assert abs(mask.mean(1).std() / np.sqrt(p * (1 - p) / cols) - 1) < 0.1
    assert abs(mask.mean(0).std() / np.sqrt(p * (1 - p) / rows) - 1) < 0.15
    for other in (dropout_mask_restated(rows, cols, p, 1234567890124, 7), dropout_mask_restated(rows, cols, p, 1234567890123, 8)):
        o = other.astype(np.float64) - other.mean()
        assert abs(corr(m, o)) < 4 / np.sqrt(n)                      # a different seed / site draws a different mask
    rq = np.arange(1 << 18, dtype=np.uint64)
    for w in rand4_restated(99, rq, 3):                              # each word uniform over its top byte
        hist = np.bincount((w >> np.uint32(24)), minlength=256)
        e = w.size / 256
        assert ((hist - e) ** 2 / e).sum() / 255 < 1.3


def test_auto_guard_slack_arithmetic():
    """AutoGuard.slack_binades (models/encoder_only.py): binades between a bound-derived f16x2 scale and the scale the
    measured maximum would have got - the number the guard compares with its threshold."""
    from protein_transformer_amd.models.encoder_only import AutoGuard

    def scale_bits(amax):                       # common.h pt_row_scale_bits: amax * scale in [2^14, 2^15)
        e = int(np.float32(amax).view(np.uint32)) >> 23
        return np.uint32(min(268 - e, 254) << 23)
    bound = 3.7
    sb = scale_bits(bound)
    got = AutoGuard.slack_binades([bound, bound / 2, bound / 300.0, bound * 2.1, 0.0, 1.0], [sb, sb, sb, sb, sb, 0x7F000000])
    #      exact bound: 0; half: 1; 1/300 (2^-8.2, mantissa of 3.7 is 1.85): 9; bound exceeded by 2.1x: -1 (or -2);
    #      all-zero operand: 0; unused atomicMin slot: 0
    assert got[0] == 0 and got[1] == 1 and got[2] in (8, 9) and got[3] < 0 and got[4] == 0 and got[5] == 0
    g = AutoGuard(2, interval=4, max_slack=8)
    assert g.want_measure() and g.off.all() and g.wide.all()           # nothing is trusted before the first measurement
    g.count_step()
    per_layer = sum(AutoGuard.PRODUCTS.values()) + len(AutoGuard.WIDE)
    assert not g.want_measure() and g.report()["fallbacks_per_step"] == 2 * per_layer
    g.off[:], g.wide[:] = False, False         # (what poll() does once a measurement with small slack / spread has landed)
    g.off[1, 1] = True                          # f1 of layer 1 off its bound: the FFN-2 forward and weight-gradient products
    g.wide[0, 3] = True                         # ff2 of layer 0 in bf16x3
    g.count_step()
    assert g.report()["fallbacks_per_step"] == (2 * per_layer + 3) / 2 and g.report()["sites_off_bounds_now"] == 1
    # spread of a weight's row scales along a contracted index: exponents 100..112 -> 12 binades; zero rows do not count
    bits = np.array([100 << 23, 112 << 23, 105 << 23, 254 << 23], dtype=np.uint32)
    assert AutoGuard.spread_binades(bits) == 12.0 and AutoGuard.spread_binades(bits[3:]) == 0.0


def test_selectors_golden(golden):
    """The two host-side selectors callers of the loss path import (structure_utils.py:19-41, losses.py:39-46) against what
    the reference returns (G12); `import protein_transformer_amd as protein_transformer` must resolve both names."""
    from protein_transformer_amd import losses
    from protein_transformer_amd.protein.structure_utils import get_backbone_from_full_coords, get_sidechain_from_full_coords
    g = golden("g12_selectors")
    assert losses.get_backbone_from_full_coords is get_backbone_from_full_coords     # losses.py:12 imports it by name
    for nd in ("2", "3"):
        crd = g["crd" + nd]
        assert np.array_equal(get_backbone_from_full_coords(crd), g["bb" + nd])
        assert np.array_equal(get_sidechain_from_full_coords(crd), g["sc" + nd])
        assert np.array_equal(get_backbone_from_full_coords(crd, invert=True), g["sc" + nd])
        t = torch.tensor(crd)
        assert np.array_equal(get_backbone_from_full_coords(t).numpy(), g["bb" + nd])
        assert np.array_equal(get_sidechain_from_full_coords(t).numpy(), g["sc" + nd])
    assert np.array_equal(g["bb2_torch"], g["bb2"])
    assert list(g["sos_ids"]) == [VOCAB.sos_id, VOCAB.eos_id]
    for i in range(int(g["n"])):
        for make in (torch.tensor, np.array, list):
            got = losses.remove_sos_eos_from_input(make(g[f"seq{i}"].tolist()))
            assert list(np.asarray(got)) == list(g[f"stripped{i}"])


def test_k_splits_of_the_small_batch_products():
    """Host-side choice of K splits (kernels.pick_split_k_rows) and the slices ptamd_gemm really makes of them
    (kernels.effective_splits = the arithmetic of build_params in csrc/gemm.hip): what the deferred reductions of round 6 rest
    on - at 2048 / 4096 tokens of the d512 model FFN-2 forward and the dX products of FFN-1 / QKV are cut 4, 4, 3 and 2, 2, 2
    ways (slices the LayerNorm kernels can sum: 2 ... 4), from 8192 tokens on nothing is split."""
    from protein_transformer_amd import kernels as K
    D, F = 512, 2048
    for tokens, want in ((2048, (4, 4, 3)), (4096, (2, 2, 2)), (8192, (1, 1, 1)), (16384, (1, 1, 1))):
        got = tuple(K.effective_splits(red, K.pick_split_k_rows(tokens, D, red)) for red in (F, F, 3 * D))
        assert got == want, (tokens, got)
    assert K.pick_split_k_rows(2048, D, D) == 1                      # wo: K = 512 is not worth cutting
    assert K.pick_split_k_rows(2048, F, D) == 1                      # FFN-1: 256 tiles already
    # slices are whole 32-blocks of K and never empty: asking for more than there are blocks gives one per block
    assert K.effective_splits(2048, 4) == 4 and K.effective_splits(1536, 4) == 4 and K.effective_splits(1536, 3) == 3
    assert K.effective_splits(96, 8) == 3 and K.effective_splits(100, 3) == 2 and K.effective_splits(32, 4) == 1
    for red in (33, 64, 1000, 1536, 2048, 4097):
        for ask in range(1, 9):
            n = K.effective_splits(red, ask)
            per = -(-(-(-red // 32)) // min(ask, -(-red // 32))) * 32     # k_per_split of build_params
            assert 1 <= n <= ask and (n - 1) * per < red <= n * per
