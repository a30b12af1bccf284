"""Is the default arithmetic (PTAMD_GEMM_AUTO: two-term f16 products on bound-derived power-of-two scales) defensible
beyond one step from a fresh initialisation?  (VERDICT of round 3, item 3.)

  * `test_guard_measures_slack`: the AutoGuard (models/encoder_only.py) measures the slack of every bound-derived scale on a
    running model, finds it small, switches nothing - and its numbers equal a direct evaluation of the operands' maxima;
  * `test_adversarial_ranges`: function-preserving rescalings that stretch the dynamic range of exactly the operands whose
    scales are bounds (LayerNorm gains spanning 2^+-8 against the columns of the next weight, FFN units spanning 2^+-10
    against the columns of layer 2, one amino acid whose embedding is 1e4 x the others): AUTO within the prediction / dRMSD
    tolerances against fp64 - or the guard fires and the step after it is;
  * `test_moved_weights_trajectory`: 200 optimizer steps at BASELINE config-2 model size on the device (AUTO) and in the
    fp64 oracle from the same initialisation: loss curves, final parameters, and a single-step parity record AT step 200;
  * `test_side_stream_is_bit_identical`: the weight-gradient products on the side stream change nothing, bit for bit.

Numbers go to gpurun_out/parity/r04_auto.json (copied to profiles/r04/ for the record).
"""
import os
import types

import numpy as np
import pytest
import torch

from parity_lib import Fp64Trainer, fp64_reference, grad_errors, params_rel_l2, update_record

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("PTAMD_AUTO_OUT", os.path.join(ROOT, "gpurun_out", "parity", "r04_auto.json"))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _setup(dev, nl, nh, dm, dff, lens, seed, dropout=0.0):
    from protein_transformer_amd import synthetic
    from protein_transformer_amd.models.encoder_only import EncoderOnlyTransformer
    from protein_transformer_amd.protein.Sequence import VOCAB
    from protein_transformer_amd.protein.Structure import nerf_forward
    L = max(lens)
    build = lambda ang, seq: nerf_forward(ang.to(dev), seq.to(dev))[0]  # noqa: E731
    batch = synthetic.make_batch(lens, L_pad=L, seed=seed, build_coords=build, frac_missing=0.02)
    torch.manual_seed(seed)
    m = EncoderOnlyTransformer(nl, nh, dm, dff, L, VOCAB, synthetic.angle_means(batch["true_ang"]), True, dropout=dropout)
    m.set_dropout(dropout)
    m = m.to(dev).train()
    with torch.no_grad():
        P = dict(m.named_parameters())
        P["output_projection.weight"].normal_(0, 0.02)
        for n, p in P.items():
            if "norm.weight" in n:
                p.add_(0.1 * torch.randn_like(p))
            elif "norm.bias" in n:
                p.add_(0.05 * torch.randn_like(p))
    return m, tuple(batch[k] for k in ("seq", "true_ang", "true_crd"))


ARGS = types.SimpleNamespace(loss="drmsd", combined_drmsd_weight=0.5, backbone_loss=False, clip=1.0)


def _one_pass(model, batch, dev, mode=None):
    """forward + loss + backward (no optimizer step) -> (pred, per-protein stats, named gradients)."""
    from protein_transformer_amd import kernels as K
    from protein_transformer_amd.losses import batch_loss
    from protein_transformer_amd.train import get_losses
    seq, ang, crd = (t.to(dev) for t in batch)
    model.gemm_mode = mode
    model.zero_grad()
    pred = model(seq, ang)
    get_losses(ARGS, pred, ang, crd, seq)
    stats = batch_loss(pred.detach(), crd, seq, do_backward=False)[0].cpu().numpy().astype(np.float64)
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    model.gemm_mode = None
    assert K.get_gemm_mode() == K.GEMM_AUTO
    return pred.detach().cpu().double(), stats, grads


def _errors(model, batch, dev, ref, mode):
    pred, stats, grads = _one_pass(model, batch, dev, mode)
    lens = [int((s != 20).sum()) for s in batch[0]]
    mask = torch.arange(batch[0].shape[1])[None, :] < torch.tensor(lens)[:, None]
    e_pred = float((pred.view_as(ref["pred"]) - ref["pred"]).abs()[mask].max())
    e_drmsd = max(abs(stats[b, 0] - ref["stats"][b][0]) / ref["stats"][b][0] for b in range(len(lens)))
    e_ln = max(abs(stats[b, 1] - ref["stats"][b][1]) for b in range(len(lens)))
    g, groups, worst = grad_errors(grads, ref["grads"])
    return {"pred_max_abs": e_pred, "drmsd_rel_max": float(e_drmsd), "lndrmsd_abs_max": float(e_ln), "grad_rel_l2": g,
            "grad_rel_l2_per_group": groups, "grad_worst_tensor": {"name": worst[0], "value": worst[1]}}


def test_guard_measures_slack(dev):
    from protein_transformer_amd import kernels as K
    from protein_transformer_amd.optim import FusedSGD
    from protein_transformer_amd.train import train_step
    model, batch = _setup(dev, 2, 8, 512, 2048, [256] * 6 + [200, 131], seed=5, dropout=0.1)
    guard = model.auto_guard
    guard.interval = 2
    opt = FusedSGD(model, lr=0.0, weight_decay=10e-3)             # (lr 0: the bound scales are the same in every step)
    data = tuple(t.to(dev) for t in batch)
    # what the guard measures, by hand: spy on the statistics launches of the measuring backward pass
    seen = []
    real = K.weight_scales

    def spy(jobs):
        if all(j.get("rows_only") and "stats" in j and "row_scale" not in j for j in jobs) and len(jobs) == 5:
            seen.append([float(j["w"].abs().max()) for j in jobs])
        return real(jobs)
    K.weight_scales = spy
    try:
        for _ in range(5):
            train_step(model, opt, ARGS, *data)
            torch.cuda.synchronize()
    finally:
        K.weight_scales = real
    rep = guard.report()
    per_layer = sum(guard.PRODUCTS.values()) + len(guard.WIDE)
    assert rep["steps"] == 5 and rep["measured_steps"] >= 2 and rep["bound_violations"] == 0
    # the first TWO steps trusted nothing (the first step's measurement is honoured by the third step's forward pass, whatever
    # the host timing): every site off its bound, every product in bf16x3; then nothing
    assert rep["fallbacks_per_step"] == 2 * 2 * per_layer / 5 and rep["sites_off_bounds_now"] == 0 and rep["products_in_bf16x3_now"] == 0
    assert guard.slack.shape == (2, 5) and (guard.slack >= 0).all() and (guard.slack <= 8).all(), guard.slack
    assert guard.spread.shape == (2, 8) and (guard.spread >= 0).all() and (guard.spread <= 6).all(), guard.spread
    # steps 0, 2, 4 measured, layers in backward order: the last completed measurement the guard has read is of step 2
    assert len(seen) == 6
    flat, _ = model.flat_parameters()
    layers = model._step_scales(flat, K.GEMM_AUTO, model.dropout, model.attn_dropout, hp=False)
    got = guard.slack
    for i in range(2):
        by_hand = np.array(seen[2 + (1 - i)])                       # step 2, layer i
        bits = np.array([int(layers[i][k][0].item()) & 0xFFFFFFFF for k in ("att_scale", "f1_scale")], dtype=np.uint32)
        want = guard.slack_binades(by_hand[:2], bits)
        assert np.array_equal(want, got[i, :2]), (i, want, got[i])
        w2 = dict(model.named_parameters())[f"encoder.enc_layers.{i}.pwff.layer2.weight"].detach()
        e = torch.frexp(w2.abs().amax(dim=0))[1]                    # exponents of the column maxima of W2
        assert guard.spread[i, 3] == float(e.max() - e.min())       # ... whose spread is what FFN-2's product is judged by
    # a model that is only ever EVALUATED takes its first measurement from a forward pass (no backward pass to measure in)
    model2, batch2 = _setup(dev, 2, 8, 512, 2048, [256] * 8, seed=6)
    model2.eval()
    g2 = model2.auto_guard
    with torch.no_grad():
        seq2 = batch2[0].to(dev)
        p1 = model2(seq2)
        assert g2.off.all() and g2.measured_steps == 0
        p2 = model2(seq2)                       # its forward waits for the measurement: bounds with small slack are trusted now
        assert g2.measured_steps == 1 and not g2.off.any() and not g2.wide.any() and (g2.slack <= 8).all()
        assert float((p1 - p2).abs().max()) < 1e-5          # two fp32-grade arithmetics
    update_record(OUT, "guard_on_a_plain_model", {"slack_binades[layer][att,f1,dz1,h1,h2]": guard.slack.tolist(),
                                                  "weight_scale_spread_binades[layer][product]": guard.spread.tolist(), **rep})


def _stretch(model, log2_gain=8, log2_unit=10, emb_factor=1e4, seed=0):
    """Function-preserving rescalings (the network computes the same function in exact arithmetic) that widen the dynamic
    range of the bound-scaled operands: LayerNorm gain j * 2^a_j with column j of the weight behind it * 2^-a_j (h1, h2);
    FFN unit n: row n of W1 and b1[n] * 2^c_n, column n of W2 * 2^-c_n (ReLU commutes with positive factors: f1, dz1);
    value channel n: row n of W_v, b_v[n] * 2^d_n, column n of W_o * 2^-d_n (att).  And - not function-preserving, but the
    same in fp64 - one amino acid's embedding row * emb_factor (a token with 1e4 x the activation norm)."""
    g = torch.Generator().manual_seed(seed)
    sd = dict(model.named_parameters())
    D = model.dmodel
    with torch.no_grad():
        for i in range(model.nlayers):
            b = f"encoder.enc_layers.{i}."
            for j, nxt in ((0, ("self_attn.wq.weight", "self_attn.wk.weight", "self_attn.wv.weight")), (1, ("pwff.layer1.weight",))):
                a = torch.exp2(torch.randint(-log2_gain, log2_gain + 1, (D,), generator=g).float()).to(sd[b + "pwff.layer1.bias"].device)
                sd[b + f"sublayer_connections.{j}.norm.weight"].mul_(a)
                sd[b + f"sublayer_connections.{j}.norm.bias"].mul_(a)
                for n in nxt:
                    sd[b + n].div_(a[None, :])
            c = torch.exp2(torch.randint(-log2_unit, log2_unit + 1, (model.dff,), generator=g).float()).to(a.device)
            sd[b + "pwff.layer1.weight"].mul_(c[:, None])
            sd[b + "pwff.layer1.bias"].mul_(c)
            sd[b + "pwff.layer2.weight"].div_(c[None, :])
            d = torch.exp2(torch.randint(-log2_unit, log2_unit + 1, (D,), generator=g).float()).to(a.device)
            sd[b + "self_attn.wv.weight"].mul_(d[:, None])
            sd[b + "self_attn.wv.bias"].mul_(d)
            sd[b + "self_attn.wo.weight"].div_(d[None, :])
        sd["encoder.input_embedding.emb.weight"][7].mul_(emb_factor)


@pytest.mark.parametrize("what", ["gains", "units", "units5", "token", "all"])
def test_adversarial_ranges(dev, what):
    """Three passes in AUTO: (1) the first pass of a model - nothing measured yet, nothing trusted; (2) the pass after the
    measurement has landed - bounds with small slack and products with a small weight-scale spread back on the fast path,
    the others where the guard put them; (3) for the record only, the round-3 behaviour (guard disabled: every bound
    trusted).  (1) and (2) must meet the prediction / dRMSD / gradient tolerances against fp64 - or, where the exact-f32
    MFMA chain itself is beyond them on these weights, be no worse than 3 x that chain."""
    from protein_transformer_amd import kernels as K
    model, batch = _setup(dev, 2, 8, 512, 2048, [256] * 7 + [173], seed=11)
    _stretch(model, log2_gain=8 if what in ("gains", "all") else 0,
             log2_unit={"units": 10, "all": 10, "units5": 5}.get(what, 0), emb_factor=1e4 if what in ("token", "all") else 1.0)
    params = {k: v.detach().cpu().double() for k, v in model.state_dict().items()}
    ref = fp64_reference(params, batch[0], batch[2], 8)
    guard = model.auto_guard
    guard.interval = 1
    rec = {"f32": _errors(model, batch, dev, ref, K.GEMM_F32), "bf16x3": _errors(model, batch, dev, ref, K.GEMM_BF16X3)}
    assert guard.train_steps == 0 and guard.off.all() and guard.wide.all()
    rec["auto_first_pass_nothing_trusted"] = _errors(model, batch, dev, ref, K.GEMM_AUTO)
    assert guard.settle()                                                                  # (due one pass later: honoured now)
    rec["auto_second_pass_guarded"] = _errors(model, batch, dev, ref, K.GEMM_AUTO)          # its forward reads the measurement
    assert guard.measured_steps >= 1 and guard.report()["bound_violations"] == 0
    rec["slack_binades[layer][att,f1,dz1,h1,h2]"] = guard.slack.tolist()
    rec["weight_scale_spread_binades[layer][" + ",".join(n for n, _ in guard.WIDE) + "]"] = guard.spread.tolist()
    rec["sites_off_bounds"], rec["products_in_bf16x3"] = int(guard.off.sum()), int(guard.wide.sum())
    guard.enabled = False
    rec["auto_with_the_guard_disabled_every_bound_trusted"] = _errors(model, batch, dev, ref, K.GEMM_AUTO)
    guard.enabled = True
    update_record(OUT, f"adversarial_{what}", rec)
    strict = rec["f32"]
    for name in ("auto_first_pass_nothing_trusted", "auto_second_pass_guarded"):
        e = rec[name]
        assert e["pred_max_abs"] < max(1e-5, 3 * strict["pred_max_abs"]), (name, e["pred_max_abs"], strict["pred_max_abs"])
        assert e["drmsd_rel_max"] < max(1e-4, 3 * strict["drmsd_rel_max"]), (name, e["drmsd_rel_max"])
        assert e["grad_rel_l2"] < max(1e-3, 3 * strict["grad_rel_l2"]), (name, e["grad_rel_l2"], strict["grad_rel_l2"])
    if what in ("gains", "all"):      # gains 2^+-8 against compensating columns: the LayerNorm-derived bounds are 2^17 .. 2^19 loose
        assert guard.off[:, :2].all() and guard.slack[:, :2].min() > 8
        assert guard.wide[:, [0, 2]].all()                 # ... and the columns of W_qkv / W_1 span 2^16
    if what in ("units", "all"):      # hidden units over 2^+-10: rows of W1 / columns of W2 span 2^20 - FFN-2 forward, dX-FFN-1
        assert guard.wide[:, [3, 6]].all() and guard.spread[:, [3, 6]].min() >= 16
    if what in ("units5", "token"):   # inside the supported range: everything stays on the fast path and meets the tolerances
        assert not guard.off.any() and not guard.wide.any()


# (round 6: these trajectories are `slow` - out of the default `-m gpu` run, whose limit on the driver's box is 1200 s: they
# took 387 s of a 976-s suite on one box and, at a fifth of the steps, still 270 s on a box with slower host cores; run them
# with --runslow / PTAMD_RUN_SLOW=1, record under profiles/.  Every single-step oracle comparison stays in the default run.)
@pytest.mark.slow
@pytest.mark.parametrize("optimizer,lr,default_steps", [("adam", 1e-4, 200), ("sgd", 1e-2, 60), ("adam", 1e-4, 40), ("sgd", 1e-2, 20)])
def test_moved_weights_trajectory(dev, optimizer, lr, default_steps):
    """BASELINE config-2 model (d256, 4 layers, 8 heads, dff 2048); 8 proteins x L <= 64 so that the fp64 oracle - a Python
    loop over the NeRF chain - makes `PTAMD_TRAJ_STEPS` (default 200 Adam / 60 SGD) steps in minutes.  AUTO resolves to
    f16x2 here (`AUTO_F16X2_MIN_WORK` lowered for the test: 512 tokens x 256 would otherwise run in bf16x3).

    What CAN be asserted about N optimizer steps.  Training this model is chaotic: the loss is a dRMSD of NeRF chains built
    from the predicted angles, Adam turns the SIGN of a gradient component at rounding level into a full-size update, and
    ReLU gates flip.  The test therefore runs, from the same initialisation and on the same batch,
      * the device in all three arithmetics (AUTO, bf16x3, exact-f32 MFMA),
      * the fp64 oracle, and the fp64 oracle once more from weights perturbed by ONE fp32 rounding (relative 6e-8) for the
        first `control` steps: what an ideal fp32 implementation could at best achieve,
    and asserts (a) the first steps agree to the single-step tolerances, (b) AUTO's distance from the fp64 trajectory is of
    the size of the exact-f32 arithmetic's and of the perturbed fp64 run's (not worse than 3 x the larger), (c) every run
    trains, (d) at the weights AUTO arrived at - they have moved by up to 35 % - the single-step errors of all three
    arithmetics ON THOSE WEIGHTS against fp64: AUTO never in another class than the strictly fp32-grade ones (3 x), and inside
    the single-step tolerances wherever that point is well-conditioned (the parity record's criterion; a trained model can
    sit where the NeRF chain amplifies any fp32 rounding), (e) the guard found every bound within 8 binades all the way."""
    from protein_transformer_amd import kernels as K
    from protein_transformer_amd.models import encoder_only as EO
    from protein_transformer_amd.optim import FusedAdam, FusedSGD
    from protein_transformer_amd.train import train_step
    steps = int(os.environ.get("PTAMD_TRAJ_STEPS", str(default_steps)))
    control = min(steps, 40)
    lens = [64, 64, 57, 64, 33, 64, 48, 64]
    modes = (("auto", K.GEMM_AUTO), ("bf16x3", K.GEMM_BF16X3), ("f32", K.GEMM_F32))
    old_min = EO.AUTO_F16X2_MIN_WORK
    EO.AUTO_F16X2_MIN_WORK = 1
    try:
        runs = {}
        for name, mode in modes:
            model, batch = _setup(dev, 4, 8, 256, 2048, lens, seed=21)
            model.gemm_mode = mode
            runs[name] = dict(model=model, curve=[], opt=(FusedAdam(model, lr=lr, betas=(0.9, 0.98), eps=1e-9, weight_decay=10e-3)
                                                          if optimizer == "adam" else FusedSGD(model, lr=lr, weight_decay=10e-3)))
        theta0 = {k: v.detach().cpu().double().clone() for k, v in runs["auto"]["model"].state_dict().items()}
        ref = Fp64Trainer(theta0, 8, optimizer=optimizer, lr=lr)
        g = torch.Generator().manual_seed(1)
        ref_p = Fp64Trainer({k: (v if k.endswith(".pe") else v * (1 + 6e-8 * (2 * torch.randint(0, 2, v.shape, generator=g) - 1)))
                             for k, v in theta0.items()}, 8, optimizer=optimizer, lr=lr)
        data = tuple(t.to(dev) for t in batch)
        curve_ref, curve_p, snap = [], [], {}
        for step in range(steps):
            for name, _ in modes:
                losses = train_step(runs[name]["model"], runs[name]["opt"], ARGS, *data)
                runs[name]["curve"].append(float(losses["drmsd-full"]))
            curve_ref.append(ref.step(batch[0], batch[2])["drmsd"])
            if step < control:
                curve_p.append(ref_p.step(batch[0], batch[2])["drmsd"])
            if step + 1 == control:        # distances from the fp64 trajectory at the control horizon
                snap = {name: params_rel_l2(runs[name]["model"].state_dict(), ref.state()) for name, _ in modes}
                snap["fp64_perturbed"] = params_rel_l2(ref_p.state(), ref.state())
        cr = np.array(curve_ref)
        moved = params_rel_l2(ref.state(), theta0)
        rec = {"model": "enc-only d256 nl4 nh8 dff2048", "lengths": lens, "optimizer": optimizer, "lr": lr, "steps": steps,
               "control_steps": control, "drmsd_first_last_fp64": [cr[0], cr[-1]], "weights_moved_rel_l2": moved,
               "parameters_rel_l2_from_the_fp64_run_at_the_control_step": snap, "runs": {}}
        for name, mode in modes:
            cd = np.array(runs[name]["curve"])
            rel = np.abs(cd - cr) / cr
            model = runs[name]["model"]
            params = {k: v.detach().cpu().double() for k, v in model.state_dict().items()}
            model.gemm_mode = None
            rec["runs"][name] = {
                "drmsd_first_last": [cd[0], cd[-1]], "loss_curve_rel_first_3_steps": rel[:3].tolist(),
                "loss_curve_rel_at_control": float(rel[control - 1]), "loss_curve_rel_max": float(rel.max()),
                "loss_curve_rel_median": float(np.median(rel)), "final_parameters_rel_l2_from_fp64": params_rel_l2(model.state_dict(), ref.state()),
                # single-step parity AT the moved weights: this run's own final weights through the device and through fp64
                "single_step_parity_at_the_last_step": _errors(model, batch, dev, fp64_reference(params, batch[0], batch[2], 8), mode)}
        relp = np.abs(np.array(curve_p) - cr[:control]) / cr[:control]
        # (d) at the weights AUTO arrived at: the three arithmetics on the SAME weights against fp64, and how well-conditioned
        # that point is (the criterion of the parity record: a 6e-8 rad perturbation of every predicted angle moves a
        # coordinate by less than 1e-3 A, no backbone bond angle within 5e-4 rad of a straight line) - a trained-for-200-steps
        # model can sit where the NeRF chain amplifies any fp32 rounding, and there no arithmetic meets the gradient tolerance
        from oracle import batched as obat
        model = runs["auto"]["model"]
        params = {k: v.detach().cpu().double() for k, v in model.state_dict().items()}
        ref_end = fp64_reference(params, batch[0], batch[2], 8)
        same = {name: _errors(model, batch, dev, ref_end, mode) for name, mode in modes}
        model.gemm_mode = None
        rad = ref_end["rad"]
        c64 = obat.generate_coords_batched(rad, batch[0], torch.float64).numpy()
        sign = torch.randint(0, 2, rad.shape, generator=g).double() * 2 - 1
        c64p = obat.generate_coords_batched(rad + 6e-8 * sign, batch[0], torch.float64).numpy()
        resp = max(float(np.abs(c64p[b, :n * 14] - c64[b, :n * 14]).max()) for b, n in enumerate(lens)) / 1e-3
        sin_bond = min(float(np.abs(np.sin(rad[b, :n, 3:6].numpy())).min()) for b, n in enumerate(lens))
        rec["at_the_weights_auto_arrived_at"] = {"errors_by_arithmetic_on_the_same_weights": same,
                                                 "coordinate_response_to_6e-8_rad_units_of_1e-3A": resp,
                                                 "smallest_abs_sin_of_a_backbone_bond_angle": sin_bond,
                                                 "well_conditioned": bool(resp <= 1.0 and sin_bond >= 5e-4)}
        rec["runs"]["fp64_perturbed_by_one_fp32_rounding"] = {"loss_curve_rel_first_3_steps": relp[:3].tolist(),
                                                               "loss_curve_rel_at_control": float(relp[-1]), "loss_curve_rel_max": float(relp.max())}
        guard = runs["auto"]["model"].auto_guard.report()
        rec["auto_guard"] = guard
    finally:
        EO.AUTO_F16X2_MIN_WORK = old_min
    long_form = steps >= (200 if optimizer == "adam" else 60)
    update_record(OUT, f"trajectory_{optimizer}" + ("" if long_form else f"_{steps}_steps"), rec)
    A, F32 = rec["runs"]["auto"], rec["runs"]["f32"]
    assert guard["bound_violations"] == 0 and guard["sites_off_bounds_now"] == 0 and guard["measured_steps"] >= steps // 16 - 1
    assert max(guard["max_slack_binades_seen"].values()) <= 8                      # (e)
    assert max(A["loss_curve_rel_first_3_steps"][:1]) < 1e-5                       # (a) the first step: the single-step tolerance
    for name, _ in modes:
        r = rec["runs"][name]
        # (c) it trains, in every arithmetic (by a tenth over the long forms; the short forms of the suite: it goes down)
        assert r["drmsd_first_last"][1] < (0.9 if long_form else 1.0) * r["drmsd_first_last"][0], name
    end = rec["at_the_weights_auto_arrived_at"]                                   # (d)
    e, strict = end["errors_by_arithmetic_on_the_same_weights"]["auto"], [end["errors_by_arithmetic_on_the_same_weights"][m] for m in ("bf16x3", "f32")]
    worst = {k: max(x[k] for x in strict) for k in ("pred_max_abs", "drmsd_rel_max", "lndrmsd_abs_max", "grad_rel_l2")}
    assert e["pred_max_abs"] < 1e-5, e                                            # the encoder alone: always
    for k, tol in (("drmsd_rel_max", 1e-4), ("lndrmsd_abs_max", 1e-6), ("grad_rel_l2", 1e-3)):
        # never in another class than the strictly fp32-grade arithmetics on the same weights ...
        assert e[k] < max(tol, 3 * worst[k]), (k, e[k], worst[k], end["well_conditioned"])
        # ... and inside the tolerance wherever the point is well-conditioned FOR THIS QUANTITY - the probe is the exact
        # arithmetics themselves: where the fma-chain fp32 product and the exact three-term split sit in a third of the
        # tolerance, so must AUTO.  (200 Adam steps can end where one backbone bond angle is within 6e-4 rad of a straight
        # line: the NeRF gradient carries 1 / sin there, the predictions agree to 1e-6 and the gradients of ALL THREE
        # arithmetics are 0.5 - 2 % off the fp64 ones, uniformly over every parameter group - profiles/r04/NOTES.md, section 2.)
        if end["well_conditioned"] and worst[k] < tol / 3:
            assert e[k] < tol, (k, e[k])
    yard = max(snap["f32"], snap["fp64_perturbed"])                                # (b)
    assert snap["auto"] < 3 * yard, snap
    assert A["final_parameters_rel_l2_from_fp64"] < 3 * max(F32["final_parameters_rel_l2_from_fp64"], rec["runs"]["bf16x3"]["final_parameters_rel_l2_from_fp64"]), rec["runs"]


def test_side_stream_is_bit_identical(dev):
    """The weight-gradient products of small batches run on a side stream (encoder_only.py: d_model >= 512, >= 4096 tokens):
    same kernels, same order per stream - the flat gradient and the updated parameters must be bit-identical to the
    single-stream pass, step after step (an ordering bug would show as a flaky mismatch)."""
    from protein_transformer_amd.optim import FusedSGD
    from protein_transformer_amd.train import train_step
    res = {}
    for side in (True, False, True):
        model, batch = _setup(dev, 2, 8, 512, 2048, [512] * 8, seed=31, dropout=0.1)
        model.side_stream_dw = side
        opt = FusedSGD(model, lr=1e-3, weight_decay=10e-3)
        data = tuple(t.to(dev) for t in batch)
        grads = []
        for _ in range(3):
            train_step(model, opt, ARGS, *data)
            model.auto_guard.settle()        # (the second step already on the trusted path: that is where the side stream matters)
            grads.append(model.flat_parameters()[1].clone())
        assert (model.__dict__.get("_side_stream") is not None) == side
        res.setdefault(side, []).append((grads, model.flat_parameters()[0].clone()))
    for grads, flat in res[True]:
        for a, b in zip(grads, res[False][0][0]):
            assert torch.equal(a, b)
        assert torch.equal(flat, res[False][0][1])


def test_stored_decisions_are_bit_identical(dev):
    """Two by-products of forward kernels replace work of backward kernels: the attention dropout decisions (keep_bits, read
    by the fused backward kernel instead of the generator) and the 1-bit gate of the FFN hidden layer (read by the gated dX
    product instead of the fp32 activation).  Same decisions either way: gradients and parameters bit-identical, step after
    step, with each of them on and off - and both really in use when on."""
    from protein_transformer_amd.optim import FusedSGD
    from protein_transformer_amd.train import train_step
    import os
    res = {}
    old = os.environ.get("PTAMD_ATTN_FUSED")
    os.environ["PTAMD_ATTN_FUSED"] = "1"          # (8 proteins x 8 heads would take the two-kernel path)
    try:
        for flags in ((True, True), (False, False), (True, False), (False, True)):
            model, batch = _setup(dev, 2, 8, 512, 2048, [512] * 8, seed=37, dropout=0.1)
            model.keep_attn_bits, model.ffn_gate_mask = flags
            opt = FusedSGD(model, lr=1e-3, weight_decay=10e-3)
            data = tuple(t.to(dev) for t in batch)
            grads = []
            for _ in range(2):
                train_step(model, opt, ARGS, *data)
                model.auto_guard.settle()     # (the second step on the trusted path: only there FFN layer 1 leaves its 1-bit gate)
                grads.append(model.flat_parameters()[1].clone())
            assert (model.__dict__.get("_attn_bits_passes", 0) > 0) == flags[0]
            assert (model.__dict__.get("_gate_mask_passes", 0) > 0) == flags[1]
            res[flags] = (grads, model.flat_parameters()[0].clone())
    finally:
        if old is None:
            os.environ.pop("PTAMD_ATTN_FUSED", None)
        else:
            os.environ["PTAMD_ATTN_FUSED"] = old
    ref = res[(False, False)]
    for flags, (grads, flat) in res.items():
        for a, b in zip(grads, ref[0]):
            assert torch.equal(a, b), flags
        assert torch.equal(flat, ref[1]), flags


def test_weight_gradient_grouping_modes(dev):
    """The weight-gradient products of a layer one by one, as an FFN pair + an attention pair, or all four as one group
    (ptamd_gemm_group): different K splits, so not bit-identical - but the same gradients to fp32 summation-order accuracy,
    each of them inside the gradient tolerance against fp64, the groups really launched, and every mode reproducible."""
    from protein_transformer_amd import kernels as K
    grads, calls = {}, {}
    real = K.linear_bwd_weight_group
    ref = None
    for mode in ("off", "pairs", "layer", "pairs"):
        model, batch = _setup(dev, 2, 8, 512, 2048, [512] * 8, seed=17)
        model.dw_group = mode
        n = []
        K.linear_bwd_weight_group = lambda jobs, sk: (n.append((len(jobs), sk)), real(jobs, sk))[1]    # noqa: E731
        try:
            _one_pass(model, batch, dev)            # nothing is trusted before the first measurement: bf16x3, no groups
            assert model.auto_guard.settle()
            assert not n
            _, _, g = _one_pass(model, batch, dev)  # bounds measured: uniform scales, f16x2, groups
            torch.cuda.synchronize()
        finally:
            K.linear_bwd_weight_group = real
        if ref is None:
            params = {k: v.detach().cpu().double() for k, v in model.state_dict().items()}
            ref = fp64_reference(params, batch[0], batch[2], 8)
        e, groups, worst = grad_errors(g, ref["grads"])
        assert e < 1e-3 and max(groups.values()) < 1e-3, (mode, e, groups, worst)
        flat = torch.cat([g[k].reshape(-1) for k in sorted(g)])
        if mode in grads:                           # the second "pairs" run: bit-identical to the first
            assert torch.equal(flat, grads[mode])
        grads[mode], calls[mode] = flat, list(n)
    assert calls["off"] == [] and calls["pairs"] == [(2, 4)] * 4 and calls["layer"] == [(4, 2)] * 2, calls
    for mode in ("pairs", "layer"):
        d = float((grads[mode] - grads["off"]).norm() / grads["off"].norm())
        assert d < 2e-6, (mode, d)


_FIVE_STEPS = r"""
import hashlib, json, sys, types
import torch
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, sys.argv[1] + "/tests")
import test_gpu_auto_guard as T
from protein_transformer_amd.optim import FusedSGD
from protein_transformer_amd.train import train_step
dev = torch.device("cuda:0")
model, batch = T._setup(dev, 2, 8, 512, 2048, [512] * 8, seed=41, dropout=0.1)
opt = FusedSGD(model, lr=1e-3, weight_decay=10e-3)
data = tuple(t.to(dev) for t in batch)
losses, switched = [], None
for i in range(5):                      # no synchronisation between the steps: the host runs ahead as far as it likes
    losses.append(float(train_step(model, opt, T.ARGS, *data)["drmsd-full"]))
    if switched is None and not model.auto_guard.off.any():
        switched = i
torch.cuda.synchronize()
flat = model.flat_parameters()[0]
print(json.dumps({"sha": hashlib.sha256(flat.cpu().numpy().tobytes()).hexdigest(), "losses": losses, "trusted_after_step": switched,
                  "guard": model.auto_guard.report()}))
"""


def test_two_fresh_processes_are_bit_identical(dev):
    """The step at which AUTO leaves the untrusting arithmetic for the bound-scaled f16x2 one is FIXED (a measurement is
    honoured two forward passes after the backward pass that took it, behind a wait for its event) - it used to depend on
    when an asynchronous copy landed.  Two fresh processes, same seeds, no synchronisation between the steps: the parameters
    after five steps are bit-identical, the losses equal, and both switched at the same step (the third)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, "-c", _FIVE_STEPS, root], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    a, b = outs
    assert a["sha"] == b["sha"] and a["losses"] == b["losses"], (a, b)
    # train_step returns after the step's forward pass has polled: the guard trusts the bounds from the THIRD step on
    assert a["trusted_after_step"] == b["trusted_after_step"] == 2, (a["trusted_after_step"], b["trusted_after_step"])
    assert a["guard"]["measured_steps"] == b["guard"]["measured_steps"] >= 1
