"""Pin the CPU oracle against the golden vectors captured from the reference.

Mirrors (and extends to NeRF / encoder / step, which the reference never tests)
/root/reference/protein_transformer/tests/test_losses.py.  CPU only.
"""
import numpy as np
import pytest
import torch
from pytest import approx

from oracle import batched, encoder, geometry, losses, step


def T(x):
    return torch.tensor(np.asarray(x))


# ---------------------------------------------------------------- G1 / G2: NeRF
def test_nerf_golden(golden):
    g = golden("g1_nerf")
    for i in range(len(g["l"])):
        d = geometry.nerf(T(g["a"][i]), T(g["b"][i]), T(g["c"][i]), T(g["l"][i]), T(g["theta"][i]), T(g["chi"][i]))
        assert np.array_equal(d.numpy(), g["d"][i])            # same ops -> bit-exact
        d2 = geometry.nerf(T(g["a"][i]), T(g["b"][i]), T(g["c"][i]), float(g["l"][i]), T(g["theta"][i]), T(g["chi"][i]))
        assert np.array_equal(d2.numpy(), g["d_pyfloat_l"][i])
    ka1 = geometry.nerf(T([0, 0, .001]).float(), T([1.442, 0, .001]).float(), T([2.0, 1.39, .001]).float(), 1.229,
                        torch.tensor(2.0944), torch.tensor(0.5))
    assert np.array_equal(ka1.numpy(), g["ka1"])
    assert ka1.numpy() == approx([1.362117, 2.308242, 0.511273], abs=2e-6)


def test_nerf_theta_range_assert():
    with pytest.raises(AssertionError):
        geometry.nerf(torch.zeros(3), torch.ones(3), torch.tensor([1., 2, 0]), 1.0, torch.tensor(3.5), torch.tensor(0.))


def test_generate_coords_golden(golden):
    g = golden("g2_coords")
    for n in range(int(g["n"])):
        s = str(g[f"seq{n}"])
        crd = geometry.generate_coords(T(g[f"ang{n}"]), T(geometry.seq_to_ids(s)))
        assert crd.shape == (len(s) * 14, 3)
        assert np.array_equal(crd.numpy(), g[f"crd{n}"]), f"case {n} ({s[:10]}..)"
    crd = geometry.generate_coords(T(g["ka2_ang"]), "GAS")
    assert np.array_equal(crd.numpy(), g["ka2_crd"])
    # SURVEY appendix F KA2 spot values
    c = crd.numpy().reshape(3, 14, 3)
    assert c[0, 2] == approx([1.98259, 1.39706, .001], abs=1e-5)
    assert c[1, 4] == approx([-.70586, .79829, -.62796], abs=1e-5)
    assert c[2, 5] == approx([.78708, 2.61326, .45568], abs=1e-5)
    assert np.all(c[0, 4:] == 0) and np.all(c[2, 6:] == 0)


def test_generate_coords_errors():
    with pytest.raises(StopIteration):
        geometry.generate_coords(torch.zeros(1, 12), torch.tensor([0]))
    with pytest.raises(KeyError):
        geometry.generate_coords(torch.zeros(3, 12), torch.tensor([0, 20, 1]))


def test_batched_matches_serial(golden):
    g = golden("g2_coords")
    picks = [n for n in range(int(g["n"])) if len(str(g[f"seq{n}"])) in (20, 64)][:6]
    for n in picks:
        s = str(g[f"seq{n}"])
        ang = T(g[f"ang{n}"])[None]
        seq = T(geometry.seq_to_ids(s))[None]
        crd32 = batched.generate_coords_batched(ang, seq, torch.float32)[0].numpy()
        crd64 = batched.generate_coords_batched(ang, seq, torch.float64)[0].numpy()
        tol = 2e-4 * max(1, len(s) / 32)
        assert np.abs(crd32 - g[f"crd{n}"]).max() < tol
        assert np.abs(crd64 - g[f"crd{n}"]).max() < tol


# ---------------------------------------------------------------- G3: dRMSD
def test_drmsd_golden(golden):
    g = golden("g3_drmsd")
    for n in range(int(g["n"])):
        d = losses.drmsd(T(g[f"a{n}"]), T(g[f"b{n}"])).item()
        assert d == float(g[f"drmsd{n}"])
        d_direct = batched.drmsd_direct(T(g[f"a{n}"]).double(), T(g[f"b{n}"]).double()).item()
        assert d_direct == approx(float(g[f"drmsd{n}"]), rel=2e-6)
    assert float(g["drmsd1"]) == approx(1.5970085859, abs=1e-6)                 # KA3
    for n in range(3):
        out = losses.pairwise_internal_dist(T(g[f"pid_in{n}"])).numpy()
        assert np.array_equal(out, g[f"pid_out{n}"])
    assert g["pid_out1"][0, 1] == approx(390.15951865)                           # test_losses.py:140-150


def test_drmsd_zero_and_permutation():
    # test_losses.py:153-174
    a = T([[0, 0, 0], [0, 1, 0], [0, 0, 2], [0, 0, 0]]).float()
    b = T([[0, 0, 2], [0, 1, 0], [0, 0, 0], [0, 0, 0]]).float()
    assert losses.drmsd(a, a) == 0
    assert losses.drmsd(a, b) != 0


# ---------------------------------------------------------------- G4: per-protein worker
def test_drmsd_work_golden(golden):
    g = golden("g4_drmsd_work")
    for b in range(4):
        r = losses.drmsd_work(g["pred_ang"][b], g["true_crd"][b], g["seq"][b])
        # forward values are bit-exact; gradients differ only by fp32 autograd accumulation order
        ref = g[f"grad{b}"]
        assert np.allclose(r[0].numpy(), ref, rtol=1e-5, atol=5e-6 * np.abs(ref).max())
        assert np.array_equal(np.array(r[1:]), g[f"vals{b}"])
    r = losses.drmsd_work(T(g["ka4_ang"]), T(g["ka4_crd"]), T(g["ka4_seq"]))
    assert np.allclose(r[1:], g["ka4_vals"], rtol=1e-6)
    assert r[1:] == approx((0.8574336, 0.0571622, 0.9411135, 0.1045682), abs=2e-6)   # KA4
    grad = r[0].numpy()
    assert np.allclose(grad, g["ka4_grad"], rtol=1e-5, atol=5e-6 * np.abs(grad).max())
    assert np.all(grad[3:] == 0)
    assert set(np.nonzero(grad[0] == 0)[0]) == {0, 3, 6, 7, 8, 9, 10, 11}
    assert set(np.nonzero(grad[2] == 0)[0]) == {2, 4, 5, 8, 9, 10, 11}


def test_batched_loss_matches_worker(golden):
    g = golden("g4_drmsd_work")
    stats, _, grad = batched.batch_loss_and_grads(T(g["pred_ang"]), T(g["seq"]), T(g["true_crd"]), torch.float64)
    for b in range(4):
        assert np.allclose(stats[b][:4], g[f"vals{b}"], rtol=2e-5)
        assert np.allclose(grad[b].numpy(), g[f"grad{b}"], rtol=2e-3, atol=2e-7)


# ---------------------------------------------------------------- G5/G6/G7: encoder, loss driver, step
def _load_sd(g, prefix="sd/"):
    sd = {k[len(prefix):]: T(v) for k, v in g.items() if k.startswith(prefix)}
    emb = sd["encoder.input_embedding.emb.weight"]
    sd["encoder.positional_enc.pe"] = encoder.positional_table(int(g["max_seq_len"]), emb.shape[1])
    return sd


def test_encoder_forward_golden(golden):
    g = golden("g567_model_step")
    sd = _load_sd(g)
    pred = encoder.encoder_forward(sd, T(g["seq"]), int(g["nhead"]))
    assert np.allclose(pred.numpy(), g["g6_pred_eval"], atol=2e-6)
    assert np.allclose(pred.numpy(), g["g6_pred_train"], atol=2e-6)


def test_compute_batch_drmsd_golden(golden):
    g = golden("g567_model_step")
    tr = step.CpuTrainer(_load_sd(g), int(g["nhead"]))
    pred = tr.forward(T(g["seq"]))
    vals = losses.compute_batch_drmsd(pred, T(g["true_crd"]), T(g["seq"]), do_backward=True)
    assert np.allclose(vals, g["g5_vals"], rtol=1e-5)
    # wk.bias gradients are mathematically 0 (softmax shift invariance) -> rounding noise only,
    # so the absolute tolerance is tied to the largest gradient of the whole model
    gmax = max(np.abs(g[k]).max() for k in g if k.startswith("g5_grad/"))
    for k, p in tr.params.items():
        ref = g["g5_grad/" + k]
        assert np.allclose(p.grad.numpy(), ref, rtol=1e-3, atol=1e-5 * gmax), k


@pytest.mark.parametrize("loss,opt", [("drmsd", "sgd"), ("combined", "sgd"), ("mse", "sgd"), ("drmsd", "adam"),
                                      ("lndrmsd", "sgd")])
def test_train_step_golden(golden, loss, opt):
    g = golden("g567_model_step")
    tag = f"g7_{loss}_{opt}"
    sd = _load_sd(g)
    tr = step.CpuTrainer(sd, int(g["nhead"]), loss=loss, optimizer=opt, lr=float(g[tag + "/lr"]))
    out = tr.step(T(g["seq"]), T(g["true_ang"]), T(g["true_crd"]))
    for k in ("loss", "drmsd-full", "lndrmsd-full", "drmsd-bb", "lndrmsd-bb", "combined-full", "mse-full", "mse-bb", "mse-sc"):
        assert float(out[k]) == approx(float(g[tag + "/loss/" + k]), rel=1e-5, abs=1e-7), k
    for k, p in tr.params.items():
        dn = float((p.detach() - sd[k]).double().norm())
        assert dn == approx(float(g[tag + "/dnorm/" + k]), rel=2e-3, abs=1e-9), k
        if tag + "/sd/" + k in g:
            assert np.allclose(p.detach().numpy(), g[tag + "/sd/" + k], rtol=1e-5, atol=1e-7), k


# ---------------------------------------------------------------- G8: angle MSE, combine, atan2
def test_mse_and_combine_golden(golden):
    g = golden("g8_mse")
    p, t = T(g["pred"]), T(g["true"])
    assert losses.mse_over_angles(p, t).item() == approx(float(g["full"]), rel=1e-6)
    assert losses.mse_over_angles(p, t, bb_only=True).item() == approx(float(g["bb"]), rel=1e-6)
    assert losses.mse_over_angles(p, t, sc_only=True).item() == approx(float(g["sc"]), rel=1e-6)
    for c, exp in zip(g["combine_in"], g["combine_out"]):
        assert losses.combine_drmsd_mse(*c) == exp
    # test_losses.py:11-19
    assert losses.combine_drmsd_mse(0.01, 0.3, 0.5, 1, 1) == 0.155
    assert losses.combine_drmsd_mse(0.02, 0.3, 1, 0.02, 1) == 1
    assert np.array_equal(losses.inverse_trig_transform(T(g["itt_in"])).numpy(), g["itt_out"])


def test_mse_loss2():
    # test_losses.py:117-120
    a = torch.zeros(8, 10, 24, dtype=torch.float64)
    assert losses.mse_over_angles(a, a - .1).item() == approx(0.01)


def test_sidechain_program_consistency():
    # mirrors tests/test_sidechains.py: programs are well-formed
    for r, prog in geometry.SC_PROGRAM.items():
        assert len(prog) <= 10
        for k, (bond, angle, tors, parents) in enumerate(prog):
            assert (parents is None) == (k == 0)
            if parents:
                assert all(p < 4 + k for p in parents)
    assert sum(len(p) for p in geometry.SC_PROGRAM.values()) == 87


# ---------------------------------------------------------------- G10: conv-enc front end
@pytest.mark.parametrize("tag", ["emb", "onehot", "linear"])
def test_conv_encoder_golden(golden, tag):
    g = golden("g10_convenc")
    pre = tag + "/sd/"
    sd = {k[len(pre):]: T(v).clone().requires_grad_() for k, v in g.items() if k.startswith(pre)}
    dl = sd["output_projection.weight"].shape[1]
    d_pe = sd["encoder.input_embedding.emb.weight"].shape[1] if tag != "onehot" else dl
    params = {**sd, "encoder.positional_enc.pe": encoder.positional_table(500, d_pe)}
    pred = encoder.encoder_forward(params, T(g["seq"]), 4, use_tanh_out=tag != "linear")   # "linear": -m conv-enc-linear-out
    assert np.allclose(pred.detach().numpy(), g[tag + "/pred"], atol=2e-6)
    if tag == "linear":      # no tanh: predictions leave [-1, 1]; the bias starts at the angle means themselves
        assert np.abs(g[tag + "/pred"]).max() > 1.0
        assert np.allclose(g["linear/init_bias"], g["angle_means"].astype(np.float32))
    (pred * T(g["w"])).sum().backward()
    gmax = max(np.abs(g[k]).max() for k in g if k.startswith(tag + "/grad/"))
    for k in g:
        if k.startswith(tag + "/grad/"):
            name = k[len(tag) + 6:]
            assert np.allclose(sd[name].grad.numpy(), g[k], rtol=1e-3, atol=1e-5 * gmax), name
