"""MI355X parity tests of the `conv-enc` front end (BASELINE config 3).

Reference: /root/reference/protein_transformer/models/convolutional_encoder.py:53-129
(`make_sequence_conv_layers` :92-104, `forward` :106-123 with the one-hot input :110-111 and the late positional
add :118-119).  Three levels:

  * the kernels of csrc/conv.hip one by one (im2col, col2im, weight pack / unpack, one-hot, positional add) and the
    Conv1d built from them (`kernels.conv1d_fwd/_bwd`) against `torch.nn.functional.conv1d` + autograd on the CPU:
    k in {3, 7, 11}, channel counts that are not multiples of 4, windows that reach across protein boundaries and
    proteins shorter than the window;
  * golden G10 (captured from the reference's ConvEncoderOnlyTransformer, embedding and one-hot variants):
    predictions abs 1e-5, every stored parameter gradient rel-L2 1e-3;
  * whole training steps of `conv-enc|3,7,11|2,2,2` at d_model 256 (`-l combined`, the loss of config 3, and
    `-l drmsd`) against the CPU oracle.
"""
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from pytest import approx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def T(x):
    return torch.tensor(np.asarray(x))


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def pad_channels(x, Cp):
    """[T, C] -> [T, Cp] zero padded (the token-major activation format of the conv kernels)."""
    out = torch.zeros(x.shape[0], Cp, dtype=x.dtype)
    out[:, :x.shape[1]] = x
    return out


# ------------------------------------------------------------------------------------------------ single kernels
@pytest.mark.parametrize("B,L,C,k", [(3, 17, 22, 3), (2, 9, 6, 7), (4, 5, 13, 11), (1, 64, 32, 5), (2, 3, 4, 7)])
def test_im2col_col2im(dev, B, L, C, k):
    from protein_transformer_amd import _lib
    from protein_transformer_amd.kernels import pad4
    lib, ptr, st = _lib.lib(), _lib.ptr, _lib.stream
    g = torch.Generator().manual_seed(B * 1000 + L * 10 + k)
    Cp, pad = pad4(C), (k - 1) // 2
    x = torch.randn(B * L, C, generator=g)
    xp = pad_channels(x, Cp).to(dev)
    col = torch.full((B * L, k * Cp), float("nan"), device=dev)
    _lib.check(lib.ptamd_im2col1d(ptr(xp), xp.stride(0), B, L, Cp, k, ptr(col), st()), "im2col")
    # reference: window slot j of token (b, l) looks at (b, l + j - pad), zero outside the protein
    ref = torch.zeros(B, L, k, Cp)
    xb = pad_channels(x, Cp).view(B, L, Cp)
    for j in range(k):
        lo, hi = max(0, pad - j), min(L, L + pad - j)
        if lo < hi:
            ref[:, lo:hi, j] = xb[:, lo + j - pad:hi + j - pad]
    assert torch.equal(col.cpu(), ref.view(B * L, k * Cp))
    # col2im is the adjoint of im2col: <col2im(d), x> == <d, im2col(x)>, and equals the explicit scatter
    d = torch.randn(B * L, k * Cp, generator=g)
    dx = torch.full((B * L, Cp), float("nan"), device=dev)
    _lib.check(lib.ptamd_col2im1d(ptr(d.to(dev)), B, L, Cp, k, ptr(dx), dx.stride(0), st()), "col2im")
    want = torch.zeros(B, L, Cp, dtype=torch.float64)
    db = d.double().view(B, L, k, Cp)
    for j in range(k):
        lo, hi = max(0, pad - j), min(L, L + pad - j)
        if lo < hi:
            want[:, lo + j - pad:hi + j - pad] += db[:, lo:hi, j]
    assert np.abs(dx.cpu().double().numpy() - want.view(B * L, Cp).numpy()).max() < 3e-5


def test_weight_pack_unpack_onehot(dev):
    from protein_transformer_amd import _lib
    from protein_transformer_amd.kernels import onehot, pad4
    lib, ptr, st = _lib.lib(), _lib.ptr, _lib.stream
    for Co, C, k in [(44, 22, 3), (8, 16, 5), (12, 5, 11)]:
        Cp = pad4(C)
        w = torch.randn(Co, C, k)
        w2 = torch.full((Co, k * Cp), float("nan"), device=dev)
        _lib.check(lib.ptamd_conv_weight_pack(ptr(w.to(dev)), Co, C, k, ptr(w2), st()), "pack")
        ref = torch.zeros(Co, k, Cp)
        ref[:, :, :C] = w.permute(0, 2, 1)
        assert torch.equal(w2.cpu(), ref.view(Co, k * Cp))
        dw = torch.randn(Co, C, k)
        acc = dw.clone().to(dev)
        _lib.check(lib.ptamd_conv_weight_unpack_add(ptr(w2), Co, C, k, ptr(acc), st()), "unpack")
        assert torch.allclose(acc.cpu(), dw + w, atol=1e-6)
    seq = torch.randint(0, 22, (3, 19))
    x = onehot(seq.to(dev), 22)
    assert x.shape == (57, 24)
    assert torch.equal(x.cpu()[:, :22], F.one_hot(seq.view(-1), 22).float()) and float(x[:, 22:].abs().sum()) == 0


def test_posenc_add(dev):
    from protein_transformer_amd.kernels import posenc_add_bwd, posenc_add_fwd
    B, L, D = 3, 21, 32
    x, pe = torch.randn(B * L, D), torch.randn(40, D)
    y = posenc_add_fwd(x.to(dev), pe.to(dev), B, L, 0.0, 5)
    ref = x + (x + pe[:L].repeat(B, 1))                      # convolutional_encoder.py:118-119, dropout off
    assert torch.allclose(y.cpu(), ref, atol=1e-6)
    dy = torch.randn(B * L, D)
    assert torch.allclose(posenc_add_bwd(dy.to(dev), 0.0, 5).cpu(), 2 * dy, atol=1e-6)
    # dropout on: the mask seen in the forward output is the one the backward pass regenerates
    p = 0.25
    y = posenc_add_fwd(x.to(dev), pe.to(dev), B, L, p, 77).cpu()
    inner = (x + pe[:L].repeat(B, 1)) / (1 - p)
    kept = (y - x).abs() > 1e-12
    assert torch.allclose(y, x + torch.where(kept, inner, torch.zeros_like(inner)), atol=1e-5)
    assert 0.65 < kept.float().mean() < 0.85
    dx = posenc_add_bwd(dy.to(dev), p, 77).cpu()
    assert torch.allclose(dx, dy * (1 + kept.float() / (1 - p)), atol=1e-5)
    y2 = posenc_add_fwd(x.to(dev), pe.to(dev), B, L, p, 78).cpu()
    assert not torch.equal((y2 - x).abs() > 1e-12, kept)     # another seed, another mask


@pytest.mark.parametrize("mode", ["f32", "auto"])
@pytest.mark.parametrize("B,L,C,Co,k", [(3, 17, 22, 44, 3), (2, 40, 32, 16, 7), (4, 9, 16, 8, 11), (2, 6, 44, 32, 11),
                                        (5, 100, 128, 64, 7)])
def test_conv1d_vs_torch(dev, B, L, C, Co, k, mode):
    """Conv1d forward, dW, db and dx against F.conv1d autograd; proteins are independent (no leakage across rows)."""
    from protein_transformer_amd import kernels as K
    old = K.get_gemm_mode()
    K.set_gemm_mode(K.GEMM_F32 if mode == "f32" else K.GEMM_AUTO)
    try:
        g = torch.Generator().manual_seed(L * 100 + C + k)
        Cp = K.pad4(C)
        x = torch.randn(B, L, C, generator=g)
        w = (torch.randn(Co, C, k, generator=g) / np.sqrt(C * k)).requires_grad_()
        b = torch.randn(Co, generator=g).requires_grad_()
        xr = x.clone().requires_grad_()
        ref = F.conv1d(xr.transpose(1, 2), w, b, padding=(k - 1) // 2).transpose(1, 2)       # [B, L, Co]
        dy = torch.randn(B, L, Co, generator=g)
        ref.backward(dy)
        xd = pad_channels(x.view(B * L, C), Cp).to(dev)
        y, w2 = K.conv1d_fwd(xd, B, L, C, w.detach().to(dev), b.detach().to(dev), k)
        tol = 2e-6 * float(ref.detach().abs().max()) * max(1.0, np.sqrt(C * k) / 8)
        assert np.abs(y.cpu().numpy() - ref.detach().reshape(B * L, Co).numpy()).max() < max(tol, 2e-6)
        dw0, db0 = torch.randn(Co, C, k, generator=g), torch.randn(Co, generator=g)        # accumulated into, not overwritten
        dw, db = dw0.clone().to(dev), db0.clone().to(dev)
        dx = K.conv1d_bwd(dy.reshape(B * L, Co).to(dev), xd, B, L, C, w2, k, dw, db, need_dx=True)
        assert rel_l2((dw.cpu() - dw0).numpy(), w.grad.numpy()) < 3e-5
        assert rel_l2((db.cpu() - db0).numpy(), b.grad.numpy()) < 3e-5
        assert rel_l2(dx.cpu().numpy()[:, :C], xr.grad.reshape(B * L, C).numpy()) < 3e-5
        assert K.conv1d_bwd(dy.reshape(B * L, Co).to(dev), xd, B, L, C, w2, k, dw, db, need_dx=False) is None
    finally:
        K.set_gemm_mode(old)


# ------------------------------------------------------------------------------------------------ golden G10
def conv_model(sd, nhead, am, kernels, reducs, use_embedding, dev, use_tanh_out=True, max_seq_len=500, dmodel=None):
    from protein_transformer_amd.models.convolutional_encoder import ConvEncoderOnlyTransformer
    from protein_transformer_amd.protein.Sequence import VOCAB
    dl = sd["output_projection.weight"].shape[1]
    dff = sd["encoder.enc_layers.0.pwff.layer1.weight"].shape[0]
    nl = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.enc_layers."))
    m = ConvEncoderOnlyTransformer(nlayers=nl, nhead=nhead, dmodel=dmodel or dl, dff=dff, max_seq_len=max_seq_len, vocab=VOCAB,
                                   angle_means=am, use_tanh_out=use_tanh_out, conv_kernel_sizes=kernels,
                                   conv_dim_reductions=reducs, use_embedding=use_embedding, conv_out_matches_dm=True,
                                   dropout=0.0)
    if not use_tanh_out:     # encoder_only.py:31-33: the bias starts at the angle means, no arctanh
        assert np.allclose(m.output_projection.bias.detach().numpy(), np.asarray(am, dtype=np.float32))
    missing, unexpected = m.load_state_dict({k: v for k, v in sd.items() if not k.endswith(".pe")}, strict=False)
    assert missing == ["encoder.positional_enc.pe"] and not unexpected       # the reference's keys, nothing else
    m.set_dropout(0.0)
    return m.to(dev)


@pytest.mark.parametrize("mode", ["f32", "auto"])
@pytest.mark.parametrize("tag", ["emb", "onehot", "linear"])
def test_conv_encoder_golden(golden, dev, tag, mode):
    """The reference's own ConvEncoderOnlyTransformer outputs and gradients (G10) through the HIP path; "linear" is
    `-m conv-enc-linear-out` (use_tanh_out=False, train.py:289-298 of the reference)."""
    from protein_transformer_amd import kernels as K
    g = golden("g10_convenc")
    pre = tag + "/sd/"
    sd = {k[len(pre):]: T(v) for k, v in g.items() if k.startswith(pre)}
    old = K.get_gemm_mode()
    K.set_gemm_mode(K.GEMM_F32 if mode == "f32" else K.GEMM_AUTO)
    try:
        m = conv_model(sd, 4, g["angle_means"], [int(k) for k in g[tag + "/kernels"]], [float(r) for r in g[tag + "/reducs"]],
                       tag != "onehot", dev, use_tanh_out=tag != "linear")
        assert set(m.state_dict().keys()) == set(sd) | {"encoder.positional_enc.pe"}
        m.train()
        m.zero_grad()
        pred = m(T(g["seq"]).to(dev))
        assert np.abs(pred.detach().cpu().numpy() - g[tag + "/pred"]).max() < 3e-5
        (pred * T(g["w"]).to(dev)).sum().backward()
        m.eval()
        with torch.no_grad():
            assert np.abs(m(T(g["seq"]).to(dev)).cpu().numpy() - g[tag + "/pred"]).max() < 3e-5
        grads = {k[len(tag) + 6:]: v for k, v in g.items() if k.startswith(tag + "/grad/")}
        gmax = max(np.abs(v).max() for v in grads.values())
        params = dict(m.named_parameters())
        assert any("conv_layers" in k for k in grads)
        for name, ref in grads.items():
            got = params[name].grad.cpu().numpy()
            if np.abs(ref).max() > 1e-4 * gmax:
                assert rel_l2(got, ref) < 1e-3, (name, rel_l2(got, ref))
            else:
                assert np.abs(got - ref).max() < 1e-5 * gmax, name
    finally:
        K.set_gemm_mode(old)


# ------------------------------------------------------------------------------------------------ whole steps
def _conv_params(nl, dm, dff, max_len, am, kernels, reducs, use_embedding, seed):
    """Reference-shaped parameter dictionary of a conv-enc model for the oracle (oracle.encoder keys + conv layers)."""
    from oracle import encoder as oenc
    torch.manual_seed(seed)
    vocab = 22
    din = dm if use_embedding else vocab
    shapes = []
    for i, (k, r) in enumerate(zip(kernels, reducs)):
        dout = dm if i == len(kernels) - 1 else int(din // r)
        shapes.append((din, dout, k))
        din = dout
    p = oenc.init_params(nl, dm, dff, max_len, am, seed=seed)
    if not use_embedding:
        del p["encoder.input_embedding.emb.weight"]
    for j, (ci, co, k) in enumerate(shapes):
        p[f"encoder.conv_layers.{j}.weight"] = torch.empty(co, ci, k).uniform_(-1, 1) * np.sqrt(3.0 / (ci * k))
        p[f"encoder.conv_layers.{j}.bias"] = torch.empty(co).uniform_(-1, 1) / np.sqrt(ci * k)
    p["output_projection.weight"].normal_(0, 0.02)
    return p


@pytest.mark.parametrize("loss,use_embedding", [("combined", True), ("drmsd", True), ("combined", False)])
def test_conv_enc_train_step_vs_oracle(dev, loss, use_embedding):
    """One full step of `-m "conv-enc|3,7,11|2,2,2" -dm 256` (2 layers to keep the CPU oracle short) with the loss of
    BASELINE config 3 (`combined` = MSE + dRMSD): losses rel 1e-4, clipped gradient norm rel 1e-3, relative L2 error
    of the whole parameter update below 5e-3."""
    from oracle import geometry, step as ostep
    from protein_transformer_amd import synthetic
    from protein_transformer_amd.optim import FusedSGD
    from protein_transformer_amd.train import train_step
    lens = [40, 23, 31, 12]
    build_cpu = lambda ang, seq: torch.stack([                                  # noqa: E731
        torch.cat([geometry.generate_coords(ang[b, :n], seq[b, :n]), torch.zeros((seq.shape[1] - n) * 14, 3)])
        for b, n in enumerate(lens)])
    batch = synthetic.make_batch(lens, L_pad=40, seed=21, build_coords=build_cpu, frac_missing=0.05)
    am = synthetic.angle_means(batch["true_ang"])
    # one-hot input: 22 -> 44 -> 44 -> 256 channels (widths must stay multiples of 4 on this path)
    kernels, reducs = [3, 7, 11], ([2.0, 2.0, 2.0] if use_embedding else [0.5, 1.0, 1.0])
    params = _conv_params(2, 256, 512, 64, am, kernels, reducs, use_embedding, seed=9)
    model = conv_model(params, 8, am, kernels, reducs, use_embedding, dev, max_seq_len=64, dmodel=256).train()
    assert model.conv_shapes == ([(256, 128, 3), (128, 64, 7), (64, 256, 11)] if use_embedding
                                 else [(22, 44, 3), (44, 44, 7), (44, 256, 11)])
    opt = FusedSGD(model, lr=1e-2, weight_decay=10e-3)
    args = types.SimpleNamespace(loss=loss, combined_drmsd_weight=0.5, backbone_loss=False, clip=1.0)
    seq, ang, crd = (batch[k] for k in ("seq", "true_ang", "true_crd"))
    losses = train_step(model, opt, args, seq.to(dev), ang.to(dev), crd.to(dev))
    ref = ostep.CpuTrainer(params, 8, loss=loss, optimizer="sgd", lr=1e-2, clip=1.0)
    ref_losses = ref.step(seq, ang, crd)
    for k in ("loss", "drmsd-full", "lndrmsd-full", "drmsd-bb", "combined-full", "mse-full", "mse-bb", "mse-sc"):
        assert float(losses[k]) == approx(float(ref_losses[k]), rel=1e-4, abs=1e-6), k
    sd = model.state_dict()
    num = den = 0.0
    for k, p in ref.params.items():
        d_ref = (p.detach() - params[k]).double()
        d_got = (sd[k].cpu() - params[k]).double()
        num += float(((d_got - d_ref) ** 2).sum())
        den += float((d_ref ** 2).sum())
        if "conv_layers" in k:
            assert float((d_got - d_ref).norm()) <= 5e-3 * float(d_ref.norm()) + 1e-7, k
    assert (num / den) ** 0.5 < 5e-3
