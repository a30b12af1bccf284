"""The parity half of BASELINE.json's metric ("dRMSD-loss delta vs ref") as NUMBERS: for every BASELINE configuration and
every arithmetic of the library (AUTO = the default of the bench line, bf16x3, exact-f32 MFMA), dropout 0, the same weights
and inputs through the HIP path and through the fp64 evaluation of the oracle's formulas:

    max |delta| of the predictions (tanh'ed cos / sin) and of the angles (atan2 output, radians, end to end),
    max |delta| of the angles given IDENTICAL encoder output (the atan2 kernel alone),
    max |delta| of the coordinates given IDENTICAL angles (Angstrom, and as a multiple of 1e-3 A * max(1, L / 128)), with
        the drift of the oracle's own fp32 NeRF from fp64 on the same angles beside it,
    per-protein drmsd (relative) and lndrmsd (absolute) deltas,
    relative L2 error of the parameter gradient - whole vector, per parameter group, and the worst single tensor.

The record is written to gpurun_out/parity/r06_parity.json (copied to profiles/r06/r06_parity.json for the judge); the test
asserts the section 8(d) tolerances on what it measured, per parameter GROUP for the gradients (a whole-vector norm cannot
see a wrong gradient in a small group: LayerNorm gains, biases).

Round 4: every configuration is run on `PTAMD_PARITY_SEEDS` independent draws (model initialisation AND batch; default 2
in the test suite, 8 in the committed record: profiles/tools/r04_parity_seeds.sh) and the record carries, per config and
arithmetic, the median and the maximum of every quantity over the draws, the per-group gradient errors, the number of
draws in which two arithmetics took different sides of a ReLU, and the SKIP RATE of ill-conditioned draws - "which
arithmetic is closest varies by draw" as a table instead of a sentence.

Round 5: two REGIMES per configuration.  "arbitrary" (rounds 3-4): a freshly initialised model with output weights
N(0, 0.02) predicts arbitrary angles - bond angles near 0 / pi, (cos, sin) pairs of length 0.02 - and most L = 1500 draws
are ill-conditioned for ANY fp32 chain (skip rate 0.7).  "realistic": the regime the reference trains in
(encoder_only.py:28-34: output weight 0, bias arctanh(angle means)) - output weights N(0, 2e-3) around the arctanh of angle
means with rotameric chi angles, so that the predicted angles stay near the truth distribution; there SURVEY 8(d) is
asserted AS WRITTEN (coordinates 1 unit of 1e-3 A * max(1, L / 128), angles 1e-4 rad unconditionally) and the conditioning
filter must fire in fewer than 10 % of the draws at L = 1500.  Both tables carry pass / fail counts against both bars.

Configs 3-5 are run on a 4-protein slice of their batch at full model size and full length: every quantity here is a
per-protein quantity (losses, coordinates) or a sum over proteins (gradient), and the fp64 oracle step on the CPU is what
bounds the run time.
"""
import json
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("PTAMD_PARITY_OUT", os.path.join(ROOT, "gpurun_out", "parity", "r06_parity.json"))
N_DRAWS = int(os.environ.get("PTAMD_PARITY_SEEDS", "2"))
N_DRAWS_REALISTIC = int(os.environ.get("PTAMD_PARITY_SEEDS_REALISTIC", "1"))
N_CONDITIONING_PROBES = int(os.environ.get("PTAMD_PARITY_PROBES", "12"))     # draws the skip rate of the realistic regime is taken over

# (BASELINE config number, model string, d_model, layers, heads, d_ff, lengths of the proteins run here, loss)
CASES = [
    (1, "enc-only", 64, 2, 8, 2048, [64, 31, 17, 48], "drmsd"),       # (-dih at the reference's default, train.py:479)
    (2, "enc-only", 256, 4, 8, 2048, [256, 256, 201, 97], "drmsd"),
    (3, "conv-enc|3,7,11|2,2,2", 256, 6, 8, 2048, [512, 512, 300, 129], "combined"),
    (4, "enc-only", 512, 6, 8, 2048, [512, 512, 411, 77], "drmsd"),
    (5, "enc-only", 512, 6, 8, 2048, [1500, 611, 1234, 200], "lndrmsd"),
]
MODES = ("auto", "bf16x3", "f32")


def _group(name):
    if "input_embedding" in name:
        return "embedding"
    if "conv_layers" in name:
        return "conv." + ("weight" if name.endswith("weight") else "bias")
    if "output_projection" in name:
        return "out." + ("weight" if name.endswith("weight") else "bias")
    if "norm" in name:
        return "layernorm." + ("gain" if name.endswith("weight") else "bias")
    if "self_attn" in name:
        return "attention." + ("weight" if name.endswith("weight") else "bias")
    return "ffn." + ("weight" if name.endswith("weight") else "bias")


def realistic_angle_means(seed=0, n=8192):
    """Mean (cos, sin) of the 12 angles for the REALISTIC regime: backbone as synthetic.sample_angles, chi angles rotameric
    (wells at -60 / 180 / +60 degrees with weights 0.5 / 0.35 / 0.15, sigma 0.25 rad - what side chains do) instead of
    uniform: a uniform chi has a mean (cos, sin) of length ~0.02, and the atan2 of a pair that short amplifies the 1e-5
    prediction tolerance to 5e-4 rad in any fp32 chain; rotameric ones give 0.3 - 0.5, as ProteinNet's do."""
    from protein_transformer_amd import synthetic
    rng = np.random.default_rng(424242 + seed)
    a = synthetic.sample_angles(rng, n).astype(np.float64)
    wells = rng.choice(np.array([-np.pi / 3, np.pi, np.pi / 3]), size=(n, 6), p=[0.5, 0.35, 0.15])
    a[:, 6:] = wells + rng.normal(0, 0.25, (n, 6))
    return np.stack([np.cos(a), np.sin(a)], -1).reshape(n, 24).mean(0)


def _make_model(dev, model, dm, nl, nh, dff, L, am, seed, out_std=0.02):
    from protein_transformer_amd.models.convolutional_encoder import ConvEncoderOnlyTransformer
    from protein_transformer_amd.models.encoder_only import EncoderOnlyTransformer
    from protein_transformer_amd.protein.Sequence import VOCAB
    torch.manual_seed(seed)
    if model.startswith("conv-enc"):
        _, ks, rs = model.split("|")
        m = ConvEncoderOnlyTransformer(nl, nh, dm, dff, L, VOCAB, am, True, [int(k) for k in ks.split(",")],
                                       [float(r) for r in rs.split(",")], True, True, dropout=0.0)
    else:
        m = EncoderOnlyTransformer(nl, nh, dm, dff, L, VOCAB, am, True, dropout=0.0)
    m.set_dropout(0.0)
    m = m.to(dev).train()
    with torch.no_grad():      # off the zero init of the output layer and off the trivial LayerNorm parameters
        P = dict(m.named_parameters())
        P["output_projection.weight"].normal_(0, out_std)
        for n, p in P.items():
            if "norm.weight" in n:
                p.add_(0.1 * torch.randn_like(p))
            elif "norm.bias" in n:
                p.add_(0.05 * torch.randn_like(p))
    return m


def _angle_delta(a, b):
    d = np.abs(a - b)
    return np.minimum(d, 2 * np.pi - d)


def _run_draw(case, draw, regime="arbitrary", probe_only=0):
    """One independent draw (model initialisation + batch) of one configuration -> its record (dict).  `regime`: see the
    module docstring.  `probe_only` = n: only the conditioning filter, on n consecutive candidate draws -> (skipped, n)."""
    from oracle import batched as obat
    from oracle import encoder as oenc
    from protein_transformer_amd import kernels as K_
    from protein_transformer_amd import synthetic
    from protein_transformer_amd.losses import angles_forward, batch_loss
    from protein_transformer_amd.protein.Structure import nerf_forward
    from protein_transformer_amd.train import get_losses
    cfg, model_s, dm, nl, nh, dff, lens, loss = case
    dev = torch.device("cuda:0")
    L = max(lens)
    B = len(lens)
    build = lambda ang, seq: nerf_forward(ang.to(dev), seq.to(dev))[0]  # noqa: E731
    # the gradient the reference back-propagates is that of sum_i lndrmsd_i whatever the reported loss (SURVEY A-7); the MSE
    # term of `combined` is covered by the G7 / G8 goldens - the record uses the dRMSD path for every config
    args = types.SimpleNamespace(loss="drmsd" if loss == "combined" else loss, combined_drmsd_weight=0.5, backbone_loss=False,
                                 clip=None)
    coord_unit = np.array([1e-3 * max(1.0, n / 128) for n in lens])
    # A freshly initialised model predicts ARBITRARY angles, and now and then a bond angle lands within ~1e-3 of 0 or pi:
    # three atoms in a line, the NeRF frame of the next atom is a normalised near-zero cross product and ANY fp32 chain -
    # the reference's own included - is off by 5 - 40 times the section-8(d) unit for that protein, its gradient with it.
    # Such a draw measures the conditioning of the input, not the kernels: it is recorded (`skipped_draws`) and the next
    # seed is taken.  Criterion (implementation-independent, fp64 only): moving every angle by one fp32 rounding of an
    # O(1) angle (6e-8 rad, random sign) moves some coordinate by more than one unit of 1e-3 A * max(1, L / 128).
    # (Found with config 4, seed 104: a predicted C-N-CA angle of -4.8e-5 rad; every atom behind it was fine on the
    # device - backbone within 2e-5 A of fp64 - but the one side-chain atom built on the collinear triple was 8.9e-3 A off,
    # and the oracle's fp32 chain 1 - 4e-3 A off from there to the end of the chain.)
    skipped = []
    gen = torch.Generator().manual_seed(1234 + cfg + 7919 * draw)
    realistic = regime == "realistic"
    n_probe_skipped = 0
    # Candidates are examined in CHUNKS (round 6): the device passes of a chunk first, then the fp64 builds of all of them at
    # once in a process pool (two Python loops over the chain per candidate: at L = 1500 they were most of this test's
    # time).  The candidate taken is the first well-conditioned one in seed order - what the one-by-one loop took.
    from parity_lib import build_coords_many
    chunk = probe_only if probe_only else (4 if L >= 512 else 2 if L >= 256 else 1)
    limit = max(32, probe_only)
    chosen = None
    for first in range(0, limit, chunk):
        cands = []
        for attempt in range(first, min(first + chunk, limit)):     # (L = 1500 chains of a random-init model: most draws hold a near-straight bond angle somewhere)
            seed = 100 + cfg + 1000 * attempt + 100000 * draw + (50000 if realistic else 0)
            batch = synthetic.make_batch(lens, L_pad=L, seed=seed, build_coords=build, frac_missing=0.02)
            seq, ang, crd = (batch[k].to(dev) for k in ("seq", "true_ang", "true_crd"))
            am = realistic_angle_means(seed) if realistic else synthetic.angle_means(batch["true_ang"])
            model = _make_model(dev, model_s, dm, nl, nh, dff, L, am, seed=7 + cfg + attempt + 131 * draw,
                                out_std=2e-3 if realistic else 0.02)
            rad_probe = angles_forward(model(seq, ang).detach()).cpu().double()
            # three backbone atoms within 5e-4 rad of a straight line (angles 3..5 = N-CA-C, CA-C-N, C-N-CA): the direction of the
            # 1e-4 A component that defines the next frame is then at the mercy of the 1e-7 A rounding of the coordinates
            # (checked first: it needs no fp64 build of the chain)
            sin_bond = np.array([np.abs(np.sin(rad_probe[b, :n, 3:6].numpy())).min() for b, n in enumerate(lens)])
            straight = sin_bond.min() < 5e-4
            sign = None if straight else torch.randint(0, 2, rad_probe.shape, generator=gen).double() * 2 - 1
            cands.append(dict(seed=seed, seq=seq, ang=ang, crd=crd, model=model, rad=rad_probe, sin_bond=sin_bond, straight=straight,
                              sign=sign))
            if not probe_only and not straight and chunk == 1:
                break
        jobs = []
        for c in cands:
            if not c["straight"]:
                jobs += [(c["rad"], c["seq"], torch.float64), (c["rad"] + 6e-8 * c["sign"], c["seq"], torch.float64)]
        built = iter(build_coords_many(jobs)) if jobs else iter(())
        for c in cands:
            if c["straight"]:
                skipped.append({"seed": c["seed"], "smallest_abs_sin_of_a_backbone_bond_angle": [float(x) for x in c["sin_bond"]]})
                n_probe_skipped += 1
                continue
            c64, c64p = next(built), next(built)
            resp = np.array([np.abs(c64p[b, :n * 14] - c64[b, :n * 14]).max() for b, n in enumerate(lens)]) / coord_unit
            if probe_only:
                n_probe_skipped += int(resp.max() > 1.0)
                continue
            if resp.max() <= 1.0:
                chosen = c
                break
            skipped.append({"seed": c["seed"], "response_to_6e-8_rad_on_every_angle_units": [float(x) for x in resp],
                            "smallest_abs_sin_of_a_backbone_bond_angle": [float(x) for x in c["sin_bond"]]})
        if probe_only:
            return n_probe_skipped, probe_only
        if chosen is not None:
            break
    if chosen is None:
        pytest.fail("no well-conditioned draw in 32 seeds")
    seed, seq, ang, crd, model = (chosen[k] for k in ("seed", "seq", "ang", "crd", "model"))
    del cands

    # ---- fp64 on the CPU: the oracle's formulas end to end
    params = {k: v.detach().cpu().double() for k, v in model.state_dict().items()}
    pe_keys = [k for k in params if k.endswith(".pe")]
    leaf = {k: v.clone().requires_grad_() for k, v in params.items() if k not in pe_keys}
    pred64 = oenc.encoder_forward({**leaf, **{k: params[k] for k in pe_keys}}, seq.cpu(), nh)
    cs = pred64.view(B, L, 12, 2)
    rad64 = torch.atan2(cs[..., 1], cs[..., 0])
    radius64 = torch.sqrt(cs[..., 1] ** 2 + cs[..., 0] ** 2).detach().numpy()     # length of the predicted (cos, sin) pair
    stats64, crd64, dang64 = obat.batch_loss_and_grads(rad64, seq.cpu(), crd.cpu(), dtype=torch.float64)
    rad64.backward(dang64)
    ref = {n: v.grad for n, v in leaf.items()}
    mask = (np.arange(L)[None, :] < np.asarray(lens)[:, None])

    rec = {"config": cfg, "model": f"{model_s} d_model={dm} n_layers={nl} n_head={nh} d_ff={dff}", "lengths": lens,
           "loss": loss, "dropout": 0.0, "reference": "fp64 evaluation of the oracle (oracle.encoder + oracle.batched)",
           "seed": seed, "draw": draw, "skipped_draws": skipped, "modes": {}, "regime": regime,
           "output_weight_std": 2e-3 if realistic else 0.02,
           "smallest_radius_of_a_predicted_cos_sin_pair": float(np.where(mask[:, :, None], radius64, np.inf).min())}
    old = K_.get_gemm_mode()
    # AUTO is recorded as the arithmetic the BENCH workloads of configs 2-5 run in (f16x2 with the guard): the 4-protein slices
    # here are below the tokens x d_model threshold at which AUTO leaves bf16x3 (config 1's real workload is below it too)
    from protein_transformer_amd.models import encoder_only as EO
    old_min = EO.AUTO_F16X2_MIN_WORK
    if cfg >= 2:
        EO.AUTO_F16X2_MIN_WORK = 1
    # The ReLU of the FFN is not differentiable at 0: an element of the hidden layer within rounding of 0 is "on" in one
    # arithmetic and "off" in another (or in fp64), and its whole back-propagated term then differs - with per-token
    # gradients spanning six decades one such element on a dominant token moves the FFN-layer-1 bias gradient and the
    # LayerNorm gradients of that layer by 1e-3 .. 1e-2 (seen in every arithmetic, the exact-f32 one included).  The gates
    # each arithmetic used (the saved hidden layer > 0, as handed to the dX product) are recorded and compared.
    gates = {}
    real_linear_bwd_input = K_.linear_bwd_input       # (the models call it through the module: patched there)

    def spy_gates(store):
        def spy(dy, w, out=None, **kw):
            if kw.get("gate") is not None:
                store.append((kw["gate"] > 0).clone())
            return real_linear_bwd_input(dy, w, out=out, **kw)
        return spy
    try:
        for mode in MODES:
            K_.set_gemm_mode({"auto": K_.GEMM_AUTO, "bf16x3": K_.GEMM_BF16X3, "f32": K_.GEMM_F32}[mode])
            if mode == "auto":
                # the steady state of AUTO is what the bench line runs: a first pass measures the slack of the bound-derived
                # scales (nothing is trusted before, models/encoder_only.py AutoGuard), the recorded pass runs on what it found
                model.auto_guard.interval = 1
                model.zero_grad()
                get_losses(args, model(seq, ang), ang, crd, seq)
                model.auto_guard.settle()
            model.zero_grad()
            pred = model(seq, ang)
            gates[mode] = []
            K_.linear_bwd_input = spy_gates(gates[mode])
            try:
                get_losses(args, pred, ang, crd, seq)
            finally:
                K_.linear_bwd_input = real_linear_bwd_input
            stats_dev, _, _ = batch_loss(pred.detach(), crd, seq, do_backward=False)
            stats_dev = stats_dev.cpu().numpy().astype(np.float64)
            p_np = pred.detach().cpu().numpy().astype(np.float64)
            rad_dev = angles_forward(pred.detach())
            rad_np = rad_dev.cpu().numpy().astype(np.float64)
            # atan2 kernel alone: fp64 atan2 of the DEVICE's encoder output
            pc = torch.from_numpy(p_np).view(B, L, 12, 2)
            rad_same = torch.atan2(pc[..., 1], pc[..., 0]).numpy()
            # NeRF alone: fp64 (and the oracle's fp32) build of the DEVICE's angles
            crd_dev = nerf_forward(rad_dev, seq)[0].cpu().numpy().astype(np.float64).reshape(B, L * 14, 3)
            crd_same64, crd_same32 = build_coords_many([(rad_dev, seq, torch.float64), (rad_dev, seq, torch.float32)])
            dcrd = np.array([np.abs(crd_dev[b, :n * 14] - crd_same64[b, :n * 14]).max() for b, n in enumerate(lens)])
            dcrd32 = np.array([np.abs(crd_same32[b, :n * 14] - crd_same64[b, :n * 14]).max() for b, n in enumerate(lens)])
            d_drmsd = np.array([abs(stats_dev[b, 0] - stats64[b][0]) / stats64[b][0] for b in range(B)])
            d_ln = np.array([abs(stats_dev[b, 1] - stats64[b][1]) for b in range(B)])
            d_bb = np.array([abs(stats_dev[b, 2] - stats64[b][2]) / stats64[b][2] for b in range(B)])
            # gradients
            got = {n: p.grad.detach().cpu().double() for n, p in model.named_parameters()}
            num = sum(float(((got[n] - ref[n]) ** 2).sum()) for n in ref)
            den = sum(float((ref[n] ** 2).sum()) for n in ref)
            groups, worst = {}, ("", 0.0)
            for n in ref:
                g = groups.setdefault(_group(n), [0.0, 0.0])
                e2, r2 = float(((got[n] - ref[n]) ** 2).sum()), float((ref[n] ** 2).sum())
                g[0] += e2
                g[1] += r2
                if r2 > 1e-24 * den and (e2 / r2) ** 0.5 > worst[1]:
                    worst = (n, (e2 / r2) ** 0.5)
            grp = {k: (v[0] / v[1]) ** 0.5 for k, v in groups.items() if v[1] > 0}
            m = {
                "pred_max_abs": float(np.abs(p_np - pred64.detach().numpy())[mask].max()),
                "angle_max_abs_rad_end_to_end": float(_angle_delta(rad_np, rad64.detach().numpy())[mask].max()),
                "angle_max_abs_rad_end_to_end_where_radius_ge_0.1": float(
                    np.where(radius64 >= 0.1, _angle_delta(rad_np, rad64.detach().numpy()), 0.0)[mask].max()),
                "angle_error_times_radius_max": float((_angle_delta(rad_np, rad64.detach().numpy()) * radius64)[mask].max()),
                "angle_max_abs_rad_given_identical_encoder_output": float(_angle_delta(rad_np, rad_same)[mask].max()),
                "coord_max_abs_A_given_identical_angles": [float(x) for x in dcrd],
                "coord_over_1e-3A_times_max(1,L/128)": [float(x) for x in dcrd / coord_unit],
                "coord_oracle_fp32_over_same_unit": [float(x) for x in dcrd32 / coord_unit],
                "drmsd_rel": [float(x) for x in d_drmsd],
                "lndrmsd_abs": [float(x) for x in d_ln],
                "drmsd_bb_rel": [float(x) for x in d_bb],
                "grad_rel_l2": (num / den) ** 0.5,
                "grad_rel_l2_per_group": grp,
                "grad_rel_l2_worst_tensor": {"name": worst[0], "value": worst[1]},
            }
            rec["modes"][mode] = m
            if mode == "auto":
                g = model.auto_guard
                rec["auto_guard"] = {"max_slack_binades": None if g.slack is None else float(g.slack.max()),
                                     "max_weight_scale_spread_binades": None if g.spread is None else float(g.spread.max()),
                                     "sites_off_bounds": int(g.off.sum()), "products_in_bf16x3": int(g.wide.sum()),
                                     "measured": g.measured_steps}
    finally:
        K_.set_gemm_mode(old)
        EO.AUTO_F16X2_MIN_WORK = old_min

    # gates in reverse layer order (the backward pass): elements whose ReLU state differs between two arithmetics
    flips = {f"{a}_vs_{b}": [int((x != y).sum().item()) for x, y in zip(gates[a], gates[b])][::-1]
             for a, b in (("auto", "bf16x3"), ("auto", "f32"), ("bf16x3", "f32"))}
    rec["relu_gate_differences_per_layer"] = flips
    return rec


def _aggregate(draws):
    """median / max over the draws of every scalar of the per-mode records (lists: their maximum per draw first)."""
    out = {}
    for mode in MODES:
        ms = [d["modes"][mode] for d in draws]
        agg = {}
        for k, v in ms[0].items():
            if isinstance(v, dict) and "value" in v:
                vals = [m[k]["value"] for m in ms]
            elif isinstance(v, dict):
                agg[k] = {g: {"median": float(np.median([m[k][g] for m in ms if g in m[k]])),
                              "max": float(np.max([m[k][g] for m in ms if g in m[k]]))} for g in v}
                continue
            elif isinstance(v, list):
                vals = [max(m[k]) for m in ms]
            else:
                vals = [m[k] for m in ms]
            agg[k] = {"median": float(np.median(vals)), "max": float(np.max(vals))}
        out[mode] = agg
    return out


def _bars(draws):
    """Pass / fail counts of (draw, arithmetic) pairs against the two coordinate / angle bars: SURVEY 8(d) as written
    (coordinates within ONE unit of 1e-3 A * max(1, L / 128) for every protein, angles within 1e-4 rad unconditionally) and
    the relaxed bar of rounds 3-4 (2 units or 3 x the drift of the reference's own fp32 chain; 1e-4 rad where the predicted
    (cos, sin) pair is >= 0.1 long)."""
    out = {}
    for m in MODES:
        c_strict = c_relaxed = a_strict = a_relaxed = 0
        for d in draws:
            r = d["modes"][m]
            dev_u, ref_u = r["coord_over_1e-3A_times_max(1,L/128)"], r["coord_oracle_fp32_over_same_unit"]
            c_strict += all(u < 1.0 for u in dev_u)
            c_relaxed += all(u < max(2.0, 3.0 * q) for u, q in zip(dev_u, ref_u))
            a_strict += r["angle_max_abs_rad_end_to_end"] < 1e-4
            a_relaxed += r["angle_max_abs_rad_end_to_end_where_radius_ge_0.1"] < 1e-4 and r["angle_error_times_radius_max"] < 1.5e-5
        n = len(draws)
        out[m] = {"draws": n, "coords_survey_8d_as_written_pass": int(c_strict), "coords_relaxed_pass": int(c_relaxed),
                  "angles_survey_8d_as_written_pass": int(a_strict), "angles_relaxed_pass": int(a_relaxed)}
    return out


def _cases_by_regime():
    """(case, regime) pairs; config 5 in the ARBITRARY regime is `slow` (round 6): 72 % of its L = 1500 draws are ill-conditioned for
    any fp32 chain and skipped - the test spends its time filtering (74 s on a fast host, 270 s on a slow one) - while the realistic
    regime, asserted as SURVEY 8(d) is written, stays in the default run; the committed record (parity_seeds.sh) has both."""
    out = []
    for c in CASES:
        for regime in ("arbitrary", "realistic"):
            marks = [pytest.mark.slow] if (c[0] == 5 and regime == "arbitrary" and "PTAMD_PARITY_SEEDS" not in os.environ) else []
            out.append(pytest.param(c, regime, id=f"{regime}-config{c[0]}", marks=marks))
    return out


@pytest.mark.parametrize("case,regime", _cases_by_regime())
def test_parity_record(case, regime):
    cfg = case[0]
    realistic = regime == "realistic"
    # (config 5 - fp64 pair sums over 21 000 atom slots, Python loops over 1500-residue chains - is a third of the suite's time
    # on a box with slow host cores: one arbitrary draw and half the conditioning probes there unless the environment asks for
    # more; the committed record is made with 8 / 4 draws and 12 probes, profiles/tools/parity_seeds.sh)
    n_arbitrary = N_DRAWS if (cfg < 5 or "PTAMD_PARITY_SEEDS" in os.environ) else 1
    draws = [_run_draw(case, d, regime) for d in range(N_DRAWS_REALISTIC if realistic else n_arbitrary)]
    n_skipped = sum(len(d["skipped_draws"]) for d in draws)
    flip_draws = {k: sum(1 for d in draws if sum(d["relu_gate_differences_per_layer"][k]) > 0)
                  for k in draws[0]["relu_gate_differences_per_layer"]}
    # which arithmetic is closest to fp64 in the whole-vector gradient, draw by draw
    closest = {m: 0 for m in MODES}
    for d in draws:
        closest[min(MODES, key=lambda m: d["modes"][m]["grad_rel_l2"])] += 1
    rec = {"config": cfg, "regime": regime, "model": draws[0]["model"], "lengths": draws[0]["lengths"], "loss": draws[0]["loss"],
           "dropout": 0.0, "reference": draws[0]["reference"], "draws": len(draws),
           "ill_conditioned_draws_skipped": n_skipped, "skip_rate": n_skipped / (n_skipped + len(draws)),
           "summary_over_draws": _aggregate(draws), "draws_with_relu_gate_differences": flip_draws,
           "bars": _bars(draws),
           "proteins_beyond_2_units": {m: sum(sum(1 for u in d["modes"][m]["coord_over_1e-3A_times_max(1,L/128)"] if u >= 2.0) for d in draws)
                                       for m in MODES},
           "auto_guard_per_draw": [d.get("auto_guard") for d in draws],
           "draws_in_which_the_arithmetic_is_closest_to_fp64_in_grad_rel_l2": closest, "per_draw": draws}
    if realistic:
        # the conditioning filter over a fixed number of candidate draws (cheap: a forward pass and two fp64 builds each): in
        # the regime the reference trains in it must almost never fire, also at L = 1500
        probes = N_CONDITIONING_PROBES if (cfg < 5 or "PTAMD_PARITY_PROBES" in os.environ) else max(4, N_CONDITIONING_PROBES // 2)
        sk, n = _run_draw(case, 1000, regime, probe_only=probes)
        rec["conditioning_probe"] = {"candidate_draws": n, "ill_conditioned": sk, "skip_rate": sk / n}
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    allrec = {}
    if os.path.exists(OUT):
        with open(OUT) as f:
            allrec = json.load(f)
    allrec[f"config{cfg}" + ("_realistic" if realistic else "")] = rec
    with open(OUT, "w") as f:
        json.dump(allrec, f, indent=1, sort_keys=True)
    print(json.dumps(rec["summary_over_draws"]["auto"], indent=1))
    print(json.dumps(rec["bars"]))
    for d in draws:
        _assert_draw(d)
    if realistic:
        assert rec["conditioning_probe"]["skip_rate"] < 0.1, rec["conditioning_probe"]
        assert n_skipped == 0 or rec["skip_rate"] < 0.5, rec["skip_rate"]


def _assert_draw(rec):
    flips = rec["relu_gate_differences_per_layer"]
    realistic = rec.get("regime") == "realistic"
    # ---- SURVEY 8(d) tolerances on the measured numbers, every arithmetic
    for mode, m in rec["modes"].items():
        assert m["pred_max_abs"] < 1e-5, (mode, m["pred_max_abs"])
        assert m["angle_max_abs_rad_given_identical_encoder_output"] < 1e-6, mode
        if realistic:
            # the regime the reference trains in: SURVEY 8(d) AS WRITTEN - angles within 1e-4 rad whatever the length of the
            # predicted pair, coordinates of every protein within ONE unit of 1e-3 A * max(1, L / 128) of the fp64 build
            assert m["angle_max_abs_rad_end_to_end"] < 1e-4, (mode, m["angle_max_abs_rad_end_to_end"])
            assert all(u < 1.0 for u in m["coord_over_1e-3A_times_max(1,L/128)"]), (mode, m["coord_over_1e-3A_times_max(1,L/128)"])
        # An angle is atan2 of a predicted (cos, sin) pair of length r: a prediction error e (tolerance 1e-5) turns the angle
        # by e / r.  1e-4 rad is asserted where r >= 0.1 (where the prediction tolerance implies it); a random-init model also
        # predicts pairs of length 0.02 - 0.03, whose angle moves by 1e-4 under the fp32 rounding of ANY chain (recorded:
        # `angle_max_abs_rad_end_to_end`) - for those the bound is the prediction tolerance itself, error x r < 1.5e-5.
        assert m["angle_max_abs_rad_end_to_end_where_radius_ge_0.1"] < 1e-4, mode
        assert m["angle_error_times_radius_max"] < 1.5e-5, mode
        # Coordinates given identical angles, against the fp64 build of the same angles.  SURVEY 8(d) quotes 1e-3 A * L / 128
        # as the drift it MEASURED between two fp32 chains on realistic angles; on the arbitrary angles of a freshly
        # initialised model the oracle's own fp32 chain (pinned bit-exact to the reference) is 0.3 - 5 such units from
        # fp64 (`coord_oracle_fp32_over_same_unit`), the device path 0.01 - 1.6 (profiles/r03_parity.json).  Asserted: never
        # beyond 2 units (the tolerance of tests/test_gpu_loss_path.py), and not worse than the fp32 chain of the
        # reference's formulas by more than the spread between two such chains.
        # (Round 4, with several draws per configuration: a protein now and then passes the conditioning filter above and is
        # still 2 - 4 units off - on the device AND in the reference's own fp32 chain.  The cap of 2 units holds where the
        # reference's chain is itself within 2/3 of a unit; beyond that the device must not be worse than 3 x that chain.
        # `proteins_beyond_2_units` in the record counts how often the second clause was needed.)
        dev_u, ref_u = m["coord_over_1e-3A_times_max(1,L/128)"], m["coord_oracle_fp32_over_same_unit"]
        assert all(d < max(2.0, 3.0 * r) for d, r in zip(dev_u, ref_u)), (mode, dev_u, ref_u)
        assert max(m["drmsd_rel"]) < 1e-4 and max(m["drmsd_bb_rel"]) < 1e-4, (mode, m["drmsd_rel"])
        assert max(m["lndrmsd_abs"]) < 1e-6, (mode, m["lndrmsd_abs"])
        # Gradients: rel-L2 1e-3 on the whole vector (section 8(d)); per parameter group 2e-3.  What is measured here is
        # mostly NOT the arithmetic of the backward pass: a 1e-6 difference in the predictions moves the NeRF chain and
        # flips ReLU gates of units whose pre-activation is within rounding of zero (the worst tensor is always a
        # pwff.layer1.weight), the same for all three arithmetics - see the exact-f32 column of the record.
        assert m["grad_rel_l2"] < 1e-3, (mode, m["grad_rel_l2"])
        for gname, e in m["grad_rel_l2_per_group"].items():
            assert e < 2e-3, (mode, gname, e)
    # the default arithmetic must not be in a different class from the strictly fp32-grade ones in ANY parameter group
    # (comparable only when AUTO took the same side of every ReLU as one of them: otherwise the difference is the kink's,
    # the absolute bounds above still hold, and the record shows which layers differ)
    auto = rec["modes"]["auto"]["grad_rel_l2_per_group"]
    same_gates = [b for b in ("bf16x3", "f32") if sum(flips[f"auto_vs_{b}"]) == 0]
    for gname in auto:
        if not same_gates:
            break
        strict = max(rec["modes"][b]["grad_rel_l2_per_group"][gname] for b in same_gates)
        assert auto[gname] < max(3.0 * strict, 2e-4), (gname, auto[gname], strict, same_gates)
