"""MI355X unit parity tests of the encoder / optimizer kernels against plain PyTorch-CPU fp32/fp64 math.

Each HIP kernel is called through the C ABI (protein_transformer_amd.kernels -> libptamd.so) and
compared with the torch op the reference uses (torch.nn.Linear, LayerNorm, softmax attention,
Embedding, SGD/Adam, clip_grad_norm_).  Tolerances are fp32 reduction-order tolerances and are
written next to each check.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def assert_close(got, ref, rtol, atol, what=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs()
    assert bool((err <= bound).all()), f"{what}: max err {err.max():.3e}, worst excess {(err - bound).max():.3e}"


# ------------------------------------------------------------------------------------------------ GEMM
GEMM_MODES = [0, 1, 2, 3]   # PTAMD_GEMM_F32 (exact f32 MFMA), _BF16X3, _BF16X3_FULL, _F16X2 (row-scaled f16 pairs); AUTO = 3 or 1


@pytest.fixture(params=GEMM_MODES, ids=["f32", "bf16x3", "bf16x3full", "f16x2"])
def gemm_mode(request):
    from protein_transformer_amd import kernels as K_
    old = K_.get_gemm_mode()
    K_.set_gemm_mode(request.param)
    yield request.param
    K_.set_gemm_mode(old)


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (200, 24, 520), (16384 // 8, 512, 512), (24, 512, 2048), (130, 132, 36)])
@pytest.mark.parametrize("a_km,b_km", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_layouts(dev, gemm_mode, M, N, K, a_km, b_km):
    from protein_transformer_amd import kernels as K_
    if a_km and M % 4:
        pytest.skip("k-major A needs M % 4 == 0")
    a = rnd((M, K), 1)                     # asymmetric, transpose-detecting data
    b = rnd((N, K), 2) + torch.arange(N)[:, None] * 1e-3
    ref = a.double() @ b.double().T
    A = (a.T if a_km else a).contiguous().to(dev)
    B = (b.T if b_km else b).contiguous().to(dev)
    C = torch.full((M, N), float("nan"), device=dev)
    K_.gemm(A, B, C, M=M, N=N, K=K, lda=A.stride(0), ldb=B.stride(0), ldc=N, a_kmajor=a_km, b_kmajor=b_km)
    # fp32 k-ordered fma chain: |err| <= K * 2^-24 * sum|a||b| worst case; 1e-6 * sum|a||b| covers K <= 2048
    # with a wide margin over the random-walk growth actually seen.  The split-bf16 modes must meet the same bound.
    bound = 1e-6 * (a.abs().double() @ b.abs().double().T)
    assert bool(((C.cpu().double() - ref).abs() <= bound + 1e-6).all())


def test_gemm_epilogues_and_split(dev, gemm_mode):
    from protein_transformer_amd import kernels as K_
    T, Kd, N = 384, 512, 200
    x, w, b, r = rnd((T, Kd), 3), rnd((N, Kd), 4, 0.1), rnd((N,), 5), rnd((T, N), 6)
    xd, wd, bd, rd = (t.to(dev) for t in (x, w, b, r))
    lin = x.double() @ w.double().T + b.double()
    y = K_.linear_fwd(xd, wd, bd)
    assert_close(y, lin, 1e-5, 1e-5, "bias")
    y = K_.linear_fwd(xd, wd, bd, flags=K_.EPI_RELU)
    assert_close(y, lin.clamp_min(0), 1e-5, 1e-5, "relu")
    y = K_.linear_fwd(xd, wd, bd, residual=rd, ldr=N)
    assert_close(y, lin + r.double(), 1e-5, 1e-5, "residual")
    y = K_.linear_fwd(xd, wd, bd, flags=K_.EPI_TANH)
    assert_close(y, torch.tanh(lin), 1e-5, 1e-5, "tanh")
    # backward products
    dy = rnd((T, N), 7)
    dx = K_.linear_bwd_input(dy.to(dev), wd)
    assert_close(dx, dy.double() @ w.double(), 1e-5, 1e-5, "dX")
    # ... fused with the backward of a ReLU + dropout layer whose saved output is `gate`
    gate = (rnd((T, Kd), 72) * (rnd((T, Kd), 73) > -0.6)).clamp_min(0)
    dxg = K_.linear_bwd_input(dy.to(dev), wd, gate=gate.to(dev), gate_dropout_p=0.2)
    assert_close(dxg, (dy.double() @ w.double()) * (gate > 0).double() / 0.8, 1e-5, 1e-5, "dX through the ReLU/dropout gate")
    dw = torch.ones(N, Kd, device=dev)
    K_.linear_bwd_weight(dy.to(dev), xd, dw)
    assert_close(dw, 1 + dy.double().T @ x.double(), 1e-5, 2e-5, "dW accumulate")
    # bias gradient fused into the dW product (column sums of dy), unsplit and split-K
    for rows in (384, 4096):
        dyl, xl = rnd((rows, N), 70), rnd((rows, Kd), 71)
        dw, db = torch.zeros(N, Kd, device=dev), torch.full((N,), 2.0, device=dev)
        K_.linear_bwd_weight(dyl.to(dev), xl.to(dev), dw, db)
        assert_close(dw, dyl.double().T @ xl.double(), 1e-5, 1e-4, "dW with fused colsum")
        assert_close(db, 2 + dyl.double().sum(0), 1e-5, 1e-4, "fused bias gradient")
    # explicit split-K equals the unsplit product up to reassociation
    big_t = 4096
    x2, dy2 = rnd((big_t, 256), 8), rnd((big_t, 128), 9)
    dw2 = torch.zeros(128, 256, device=dev)
    K_.gemm(dy2.to(dev), x2.to(dev), dw2, M=128, N=256, K=big_t, lda=128, ldb=256, ldc=256, a_kmajor=True,
            b_kmajor=True, split_k=16)
    assert_close(dw2, dy2.double().T @ x2.double(), 1e-5, 1e-4, "split-K")
    bsum = torch.zeros(N, device=dev)
    K_.colsum(dy.to(dev), bsum, accumulate=False)
    assert_close(bsum, dy.double().sum(0), 1e-5, 1e-5, "colsum")


def test_gemm_dropout_mask_roundtrip(dev, gemm_mode):
    from protein_transformer_amd import kernels as K_
    T, Kd, N, p = 512, 64, 256, 0.1
    x, w = rnd((T, Kd), 10).to(dev), rnd((N, Kd), 11).to(dev)
    y0 = K_.linear_fwd(x, w, None)
    yd = K_.linear_fwd(x, w, None, dropout_p=p, seed=1234, stream_id=7)
    kept = yd != 0
    frac = kept.float().mean().item()
    assert abs(frac - (1 - p)) < 0.01, frac
    assert torch.allclose(yd[kept], y0[kept] / (1 - p), rtol=1e-6, atol=1e-7)
    m = K_.dropout_bwd(torch.ones(T, N, device=dev), p, 1234, 7)
    assert torch.equal(m != 0, kept)
    assert torch.allclose(m[kept], torch.full_like(m[kept], 1 / (1 - p)))
    # the device generator is the documented one (numpy restatement in tests/test_host_logic.py)
    from test_host_logic import dropout_mask_restated
    assert np.array_equal(kept.cpu().numpy(), dropout_mask_restated(T, N, p, 1234, 7))
    # a different stream id draws a different mask
    y2 = K_.linear_fwd(x, w, None, dropout_p=p, seed=1234, stream_id=8)
    assert not torch.equal(y2 != 0, kept)
    # relu + dropout backward only needs the saved output
    h = K_.linear_fwd(x, w, None, flags=K_.EPI_RELU, dropout_p=p, seed=99, stream_id=3)
    g = K_.relu_dropout_bwd(torch.ones_like(h), h, p)
    assert torch.equal(g != 0, h > 0)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_gemm_randomised(dev, seed):
    """40 random (shape, operand layout, epilogue, split-K, fused bias gradient, arithmetic mode) combinations per
    seed against fp64: every code path of ptamd_gemm meets the same bound, 2e-6 * (sum |a||b| + 1)."""
    from protein_transformer_amd import kernels as K_
    rng = np.random.default_rng(seed)
    old = K_.get_gemm_mode()
    try:
        for _ in range(40):
            mode = int(rng.integers(0, 5))
            a_km, b_km = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            M, N, Kd = (int(rng.integers(1, 700)) for _ in range(3))
            if a_km:
                M = max(4, M // 4 * 4)
            if b_km:
                N = max(4, N // 4 * 4)
            if rng.integers(0, 4) == 0:
                Kd = int(rng.choice([16, 20, 36, 512, 4096 + 16 * int(rng.integers(0, 3))]))
            if not (a_km and b_km):
                Kd = max(4, Kd // 4 * 4)
            split = int(rng.choice([1, 1, 2, 5, 16])) if Kd >= 64 else 1
            use_bias, use_res, relu, tanh_, accum = (bool(rng.integers(0, 2)) for _ in range(5))
            colsum = a_km and bool(rng.integers(0, 2))
            gate = (not use_res) and bool(rng.integers(0, 4) == 0)
            a = torch.tensor(rng.normal(0, 1, (M, Kd)), dtype=torch.float32)
            b = torch.tensor(rng.normal(0, 1, (N, Kd)), dtype=torch.float32)
            bias = torch.tensor(rng.normal(0, 1, N), dtype=torch.float32) if use_bias else None
            res = torch.tensor(rng.normal(0, 1, (M, N)), dtype=torch.float32) if (use_res or gate) else None
            c0 = torch.tensor(rng.normal(0, 1, (M, N)), dtype=torch.float32)
            cs0 = torch.tensor(rng.normal(0, 1, M), dtype=torch.float32)
            ref = a.double() @ b.double().T
            if use_bias:
                ref = ref + bias.double()
            if relu:
                ref = ref.clamp_min(0)
            if use_res:
                ref = ref + res.double()
            if gate:
                ref = torch.where(res.double() > 0, ref * 1.25, torch.zeros_like(ref))
            if tanh_:
                ref = torch.tanh(ref)
            if accum:
                ref = ref + c0.double()
            flags = ((K_.EPI_RELU if relu else 0) | (K_.EPI_TANH if tanh_ else 0) | (K_.EPI_ACCUM if accum else 0)
                     | (K_.EPI_GATE if gate else 0))
            K_.set_gemm_mode(mode)
            A = (a.T if a_km else a).contiguous().to(dev)
            B = (b.T if b_km else b).contiguous().to(dev)
            C, cs = c0.clone().to(dev), cs0.clone().to(dev)
            K_.gemm(A, B, C, M=M, N=N, K=Kd, lda=A.stride(0), ldb=B.stride(0), ldc=N, a_kmajor=a_km, b_kmajor=b_km,
                    bias=bias.to(dev) if use_bias else None, residual=res.to(dev) if res is not None else None, ldr=N,
                    flags=flags, split_k=split, colsum=cs if colsum else None, gate_scale=1.25)
            what = dict(mode=mode, M=M, N=N, K=Kd, a_km=a_km, b_km=b_km, split=split, bias=use_bias, res=use_res,
                        relu=relu, tanh=tanh_, accum=accum, colsum=colsum, gate=gate)
            scale = (a.abs().double() @ b.abs().double().T) + 1.0
            assert ((C.cpu().double() - ref).abs() / scale).max().item() < 2e-6, what
            if colsum:
                cref = cs0.double() + a.double().sum(1)
                assert ((cs.cpu().double() - cref).abs() / (a.abs().double().sum(1) + 1)).max().item() < 2e-6, what
    finally:
        K_.set_gemm_mode(old)


@pytest.mark.parametrize("mode", [1, 3])
def test_gemm_tile_heights_agree(dev, mode):
    """The staging GEMM takes 128-row tiles where 256-row tiles would leave most CUs idle (gemm_split_kernel.h, launch): the
    same rows computed as part of a product that fills the chip with 256-row tiles and as a small product of their own
    (128-row tiles) must be the same bits - epilogue and dropout masks included: an element's k order does not change."""
    from protein_transformer_amd import kernels as K_
    g = torch.Generator().manual_seed(5 + mode)
    T, Ts, N, Kd = 16384, 1024, 512, 512
    a = torch.randn(T, Kd, generator=g).to(dev)
    w, bias = (torch.randn(N, Kd, generator=g) * 0.05).to(dev), torch.randn(N, generator=g).to(dev)
    res = torch.randn(T, N, generator=g).to(dev)

    def run(rows):
        c = torch.empty(rows, N, device=dev)
        K_.gemm(a[:rows], w, c, M=rows, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, bias=bias, residual=res[:rows], ldr=N, dropout_p=0.1,
                seed=11, stream_id=4, arith=mode)
        return c
    big, small = run(T), run(Ts)
    assert torch.equal(big[:Ts], small)
    assert (small == res[:Ts]).float().mean().item() > 0.05          # (dropped elements: the residual alone)
    # round 6: three tile heights - 256 rows (16384 tokens), 128 (8192: 64-row tiles would need a second round), 64 (<= 4096)
    for rows in (8192, 4096, 2048):
        assert torch.equal(big[:rows], run(rows)), rows


def test_gemm_split_bf16_is_fp32_grade(dev):
    """The default arithmetic (three-term bf16 split, six MFMA products) against fp64, next to the exact-f32 MFMA.

    Claim stated in include/ptamd.h: the split product is at least as close to the exact result as the fp32 fma
    chain.  Checked on the shapes of the training step (K = 512, 2048, 16384-token reduction) with operands of very
    different magnitudes (activations ~1, gradients ~1e-4, weights ~0.05) and with heavy cancellation.
    """
    from protein_transformer_amd import kernels as K_
    old = K_.get_gemm_mode()
    try:
        cases = [(512, 256, 512, 1.0, 0.05), (512, 256, 2048, 1.0, 0.02), (256, 256, 16384, 1e-4, 1.0),
                 (384, 128, 4096, 3e3, 1e-6)]
        for M, N, Kd, sa, sb in cases:
            g = torch.Generator().manual_seed(M + Kd)
            a = torch.randn(M, Kd, generator=g) * sa
            b = torch.randn(N, Kd, generator=g) * sb
            a[:, ::7] *= 64.0                       # wide dynamic range inside one dot product
            b[::5] *= 1e-3
            ref = a.double() @ b.double().T
            scale = (a.abs().double() @ b.abs().double().T)        # sum |a||b|: the natural error unit
            errs = {}
            for mode in (K_.GEMM_F32, K_.GEMM_BF16X3, K_.GEMM_BF16X3_FULL):
                K_.set_gemm_mode(mode)
                c = torch.empty(M, N, device=dev)
                K_.gemm(a.to(dev), b.to(dev), c, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N)
                e = (c.cpu().double() - ref).abs() / scale
                errs[mode] = (e.max().item(), e.pow(2).mean().sqrt().item())
            f32_max, f32_rms = errs[K_.GEMM_F32]
            for mode in (K_.GEMM_BF16X3, K_.GEMM_BF16X3_FULL):
                mx, rms = errs[mode]
                assert rms <= 1.05 * f32_rms + 1e-9, (M, N, Kd, mode, errs)
                assert mx <= 1.5 * f32_max + 2e-8, (M, N, Kd, mode, errs)
                assert mx <= 2e-6, (M, N, Kd, mode, errs)          # absolute, in units of sum |a||b|
            print(f"K={Kd}: rel err (max, rms) f32 {errs[0]}, bf16x3 {errs[1]}, bf16x3full {errs[2]}")
    finally:
        K_.set_gemm_mode(old)


def test_gemm_f16x2_error_model(dev):
    """PTAMD_GEMM_F16X2 (row-scaled f16 pairs, three products) against fp64: what include/ptamd.h states.

    (1) operands whose rows span a moderate range - every tensor of the training step: activations, gradients with
        per-token magnitudes from 1e-9 to 1e3, weights - meet the fp32 bound, in units of sum |a||b|;
    (2) rows spanning more than 18 binades only meet the norm-wise bound 2^-20 sum|a||b| + 2^-36 K max|a_row| max|b_col|;
    (3) zero rows, single-element rows, huge and tiny magnitudes do not overflow, underflow or produce NaN.
    """
    from protein_transformer_amd import kernels as K_
    old = K_.get_gemm_mode()

    def run(a, b, mode, a_km=False, b_km=False, split=1):
        K_.set_gemm_mode(mode)
        M, N, Kd = a.shape[0], b.shape[0], a.shape[1]
        A = (a.T if a_km else a).contiguous().to(dev)
        B = (b.T if b_km else b).contiguous().to(dev)
        c = torch.empty(M, N, device=dev)
        K_.gemm(A, B, c, M=M, N=N, K=Kd, lda=A.stride(0), ldb=B.stride(0), ldc=N, a_kmajor=a_km, b_kmajor=b_km, split_k=split)
        return c.cpu().double()
    try:
        g = torch.Generator().manual_seed(11)
        # (1) per-row magnitudes over 12 decades, a few binades inside a row
        for (M, N, Kd, a_km, b_km, split) in [(512, 256, 512, False, False, 1), (384, 512, 2048, False, True, 1),
                                              (516, 128, 8192, True, True, 8), (260, 132, 516, True, False, 1)]:   # (k-major A of 516 x 8192: the chunked scale pass)
            a = torch.randn(M, Kd, generator=g) * torch.exp(torch.empty(M, 1).uniform_(-20, 7, generator=g))
            b = torch.randn(N, Kd, generator=g) * torch.exp(torch.empty(N, 1).uniform_(-8, 2, generator=g))
            a[:, ::7] *= 64.0
            ref, unit = a.double() @ b.double().T, a.abs().double() @ b.abs().double().T
            e16 = ((run(a, b, K_.GEMM_F16X2, a_km, b_km, split) - ref).abs() / unit)
            e32 = ((run(a, b, K_.GEMM_F32, a_km, b_km, split) - ref).abs() / unit)
            print(f"K={Kd}: rel err (max, rms) f16x2 {e16.max():.2e} {e16.pow(2).mean().sqrt():.2e}  "
                  f"f32 {e32.max():.2e} {e32.pow(2).mean().sqrt():.2e}")
            assert e16.max().item() < 1e-6, (M, N, Kd)
            assert e16.pow(2).mean().sqrt().item() < 1.5 * e32.pow(2).mean().sqrt().item() + 2e-8, (M, N, Kd)
        # (2) 40 binades inside every row: the norm-wise bound holds, the component-wise one need not
        M, N, Kd = 256, 256, 512
        a = torch.randn(M, Kd, generator=g) * torch.exp(4 * torch.randn(M, Kd, generator=g))
        b = torch.randn(N, Kd, generator=g) * torch.exp(4 * torch.randn(N, Kd, generator=g))
        ref, unit = a.double() @ b.double().T, a.abs().double() @ b.abs().double().T
        norm = 2.0 ** -20 * unit + 2.0 ** -36 * Kd * a.abs().amax(1, keepdim=True).double() * b.abs().amax(1).double()
        err = (run(a, b, K_.GEMM_F16X2) - ref).abs()
        assert bool((err <= norm).all()), (err / norm).max()
        assert (err / unit).max().item() > 1e-5          # ... and the fp32 bound is indeed missed here
        # (3) degenerate rows and extreme magnitudes
        a = torch.randn(64, 256, generator=g)
        b = torch.randn(64, 256, generator=g)
        a[0] = 0.0; a[1] = 0.0; a[1, 5] = 3.0; a[2] *= 1e25; a[3] *= 1e-30; a[4] *= 1e-42   # row 4: subnormals only
        b[0] = 0.0; b[2] *= 1e-25; b[3] *= 1e8
        ref, unit = a.double() @ b.double().T, a.abs().double() @ b.abs().double().T
        c = run(a, b, K_.GEMM_F16X2)
        assert bool(torch.isfinite(c).all())
        keep = torch.ones(64, dtype=torch.bool); keep[4] = False          # (a row of subnormals is flushed to zero)
        assert bool(((c - ref).abs() <= 1e-6 * unit + 1e-37)[keep].all())       # (1e-37: products below the f32 range)
        assert bool((c[4] == 0).all()) and bool((c[0] == 0).all()) and bool((c[:, 0] == 0).all())
    finally:
        K_.set_gemm_mode(old)


# ------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("T,D", [(1000, 512), (37, 64), (256, 256), (64, 2048)])
def test_layernorm(dev, T, D):
    from protein_transformer_amd import kernels as K_
    x = (rnd((T, D), 12) * 3 + 0.5).requires_grad_()
    g, b = (rnd((D,), 13) + 1.5).requires_grad_(), rnd((D,), 14).requires_grad_()
    y = F.layer_norm(x, (D,), g, b, 1e-5)
    dy = rnd((T, D), 15)
    y.backward(dy)
    yd, mean, rstd = K_.layernorm_fwd(x.detach().to(dev), g.detach().to(dev), b.detach().to(dev))
    assert_close(yd, y, 1e-5, 1e-5, "ln fwd")
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    dx = K_.layernorm_bwd(dy.to(dev), x.detach().to(dev), g.detach().to(dev), mean, rstd, dg, db)
    assert_close(dx, x.grad, 1e-4, 1e-5, "ln dx")
    assert_close(dg, g.grad, 1e-4, 1e-4, "ln dgamma")
    assert_close(db, b.grad, 1e-4, 1e-4, "ln dbeta")


# ------------------------------------------------------------------------------------------------ embedding
def test_embedding(dev):
    from oracle.encoder import positional_table
    from protein_transformer_amd import kernels as K_
    B, L, D = 3, 50, 64
    seq = torch.randint(0, 22, (B, L), generator=torch.Generator().manual_seed(1))
    emb, pe = rnd((22, D), 16).requires_grad_(), positional_table(64, D)[0]
    x0 = emb[seq] * np.sqrt(D)
    ref = x0 + (x0 + pe[:L])
    out = K_.embed_fwd(seq.to(dev), emb.detach().to(dev), pe.to(dev), 0.0, 0)
    assert_close(out.view(B, L, D), ref, 1e-6, 1e-6, "embed fwd")
    dout = rnd((B, L, D), 17)
    ref.backward(dout)
    demb = torch.zeros(22, D, device=dev)
    K_.embed_bwd(seq.to(dev), dout.view(B * L, D).to(dev), D, 0.0, 0, demb)
    assert_close(demb, emb.grad, 1e-5, 1e-5, "embed bwd")
    # with dropout the map emb -> out is still linear for fixed masks: <bwd(dout), delta> == <dout, fwd(emb+delta) - fwd(emb)>
    p, seed = 0.1, 77
    embd, delta = emb.detach().to(dev), rnd((22, D), 18).to(dev)
    zero_pe = torch.zeros_like(pe).to(dev)
    o1 = K_.embed_fwd(seq.to(dev), embd, pe.to(dev), p, seed)
    o2 = K_.embed_fwd(seq.to(dev), embd + delta, pe.to(dev), p, seed)
    demb = torch.zeros(22, D, device=dev)
    K_.embed_bwd(seq.to(dev), dout.view(B * L, D).to(dev), D, p, seed, demb)
    lhs = (demb.double() * delta.double()).sum().item()
    rhs = ((o2 - o1).double() * dout.view(B * L, D).to(dev).double()).sum().item()
    assert lhs == pytest.approx(rhs, rel=1e-4)
    frac_zero = (K_.embed_fwd(seq.to(dev), embd, zero_pe, p, seed) == 0).float().mean().item()
    assert abs(frac_zero - p) < 0.02          # outer dropout rate


# ------------------------------------------------------------------------------------------------ attention
def ref_attention(qkv, key_ok, H, mask_keep=None, p=0.0):
    """qkv [B,L,3D] double; returns out [B,L,D] using the reference's formulation (Attention.py:14-22,55-68)."""
    B, L, D3 = qkv.shape
    D = D3 // 3
    dk = D // H
    q, k, v = (t.reshape(B, L, H, dk).transpose(1, 2) for t in qkv.split(D, dim=-1))
    s = q @ k.transpose(-2, -1) / np.sqrt(dk)
    s = s.masked_fill(~key_ok[:, None, None, :], -np.inf)
    pr = torch.softmax(s, dim=-1)
    if mask_keep is not None:
        pr = pr * mask_keep / (1 - p)
    return (pr @ v).transpose(1, 2).reshape(B, L, D), pr


@pytest.mark.parametrize("B,L,H,dk,lens", [(2, 100, 4, 8, [100, 37]), (2, 64, 2, 16, [64, 5]), (3, 200, 8, 32, [200, 129, 64]),
                                           (2, 512, 8, 64, [512, 300]), (1, 130, 2, 64, [130])])
def test_attention_forward_backward(dev, gemm_mode, B, L, H, dk, lens):
    # gemm_mode selects the matrix arithmetic: exact f32 MFMA, or (dk = 64, 32) the split-bf16 kernels of attention_split.hip
    from protein_transformer_amd import kernels as K_
    D = H * dk
    seq = torch.full((B, L), 20, dtype=torch.int64)
    for b, n in enumerate(lens):
        seq[b, :n] = torch.randint(0, 20, (n,), generator=torch.Generator().manual_seed(b))
    qkv = (rnd((B, L, 3 * D), 20, 1.5)).double().requires_grad_()
    out, _ = ref_attention(qkv, seq != 20, H)
    dout = rnd((B, L, D), 21).double()
    out.backward(dout)
    qd = qkv.detach().float().view(B * L, 3 * D).to(dev)
    o, lse = K_.attention_fwd(qd, seq.to(dev), H, 0.0, 0, 0)
    assert_close(o.view(B, L, D), out, 1e-5, 2e-6, "attention fwd")
    dqkv = K_.attention_bwd(qd, seq.to(dev), o, dout.float().view(B * L, D).to(dev), lse, H, 0.0, 0, 0)
    ref = qkv.grad.view(B * L, 3 * D)
    scale = ref.abs().max().item()
    assert_close(dqkv, ref, 1e-4, 2e-6 * max(1.0, scale), "attention bwd")


def test_attention_randomised(dev, gemm_mode):
    """30 random (batch, heads, length, padding) cases with dk = 64 or 32 - the head sizes of the split-bf16 kernels -
    forward and backward against dense fp64 attention."""
    from protein_transformer_amd import kernels as K_
    rng = np.random.default_rng(5 + gemm_mode)
    for it in range(30):
        B, H, dk, L = int(rng.integers(1, 4)), int(rng.choice([1, 2, 4])), (64, 32)[it % 2], int(rng.integers(2, 600))
        D = H * dk
        seq = torch.full((B, L), 20, dtype=torch.int64)
        for b in range(B):
            n = int(rng.integers(1, L + 1)) if b else L
            seq[b, :n] = torch.tensor(rng.integers(0, 20, n))
        qkv = torch.tensor(rng.normal(0, 1.2, (B, L, 3 * D)), dtype=torch.float32).double().requires_grad_()
        out, _ = ref_attention(qkv, seq != 20, H)
        dout = torch.tensor(rng.normal(0, 1, (B, L, D)), dtype=torch.float64)
        out.backward(dout)
        qd = qkv.detach().float().view(B * L, 3 * D).to(dev)
        o, lse = K_.attention_fwd(qd, seq.to(dev), H, 0.0, 0, 0)
        what = f"B={B} H={H} L={L} lens={(seq != 20).sum(1).tolist()}"
        assert_close(o.view(B, L, D), out, 1e-5, 2e-6, "attention fwd " + what)
        dqkv = K_.attention_bwd(qd, seq.to(dev), o, dout.float().view(B * L, D).to(dev), lse, H, 0.0, 0, 0)
        ref = qkv.grad.view(B * L, 3 * D)
        assert_close(dqkv, ref, 1e-4, 2e-6 * max(1.0, ref.abs().max().item()), "attention bwd " + what)


@pytest.mark.parametrize("dk", [64, 32])
def test_attention_dropout_consistency(dev, gemm_mode, dk):
    """Recover the dropout mask from a forward pass with V = I, then check all three gradients against
    dense torch math that uses that mask: forward, dQ and dK/dV kernels must draw identical masks."""
    from protein_transformer_amd import kernels as K_
    B, L, H, p, seed, sid = 2, dk, 2, 0.25, 4242, 5
    D = H * dk
    seq = torch.randint(0, 20, (B, L), generator=torch.Generator().manual_seed(3))
    seq[1, L - 14:] = 20
    qkv = rnd((B, L, 3 * D), 22, 1.2)
    eye = qkv.clone()
    eye[:, :, 2 * D:] = torch.eye(L)[None].repeat(B, 1, H)          # V_h = I for every head (dk == L)
    pd, _ = K_.attention_fwd(eye.view(B * L, 3 * D).to(dev), seq.to(dev), H, p, seed, sid)
    pd = pd.view(B, L, H, dk).permute(0, 2, 1, 3).cpu().double()     # [B,H,q,key] dropped probabilities
    _, pr = ref_attention(eye.double(), seq != 20, H)
    keep = (pd != 0)
    live = pr > 1e-12
    assert abs(keep[live].float().mean().item() - (1 - p)) < 0.02
    assert torch.allclose(pd[keep], (pr / (1 - p))[keep], rtol=1e-4, atol=1e-7)
    # now real V: forward and backward with the recovered mask
    q64 = qkv.double().requires_grad_()
    out, _ = ref_attention(q64, seq != 20, H, mask_keep=keep.double(), p=p)
    dout = rnd((B, L, D), 23).double()
    out.backward(dout)
    qd = qkv.view(B * L, 3 * D).to(dev)
    o, lse = K_.attention_fwd(qd, seq.to(dev), H, p, seed, sid)
    assert_close(o.view(B, L, D), out, 1e-5, 5e-6, "attention fwd (dropout)")
    dqkv = K_.attention_bwd(qd, seq.to(dev), o, dout.float().view(B * L, D).to(dev), lse, H, p, seed, sid)
    ref = q64.grad.view(B * L, 3 * D)
    assert_close(dqkv, ref, 1e-4, 5e-6 * max(1.0, ref.abs().max().item()), "attention bwd (dropout)")


@pytest.mark.parametrize("B,L,H,dk", [(2, 512, 8, 64), (3, 300, 4, 32)])
def test_attention_f16x2_wide_row_ranges(dev, B, L, H, dk):
    """The two-term f16 kernels scale K / V / Q / dO per row (lanes) or per group of four rows (LDS tiles) and adapt the
    common power of two of the probability / dS operands on line: rows whose magnitudes span five decades (V 1e-3..10,
    K 1e-2..3, dO 1e-8..1e-4) must come out as accurately as with the exact three-term bf16 kernels (norm-wise, against
    dense fp64 attention; measured 3..6e-7 in every arithmetic - a round-2 probe, git history 160951b)."""
    from protein_transformer_amd import kernels as K_
    g = torch.Generator().manual_seed(11)
    D = H * dk
    qkv = torch.randn(B, L, 3 * D, generator=g, dtype=torch.float64)
    dout = torch.randn(B, L, D, generator=g, dtype=torch.float64)
    qkv[:, :, D:2 * D] *= 10 ** (torch.rand(B, L, 1, generator=g, dtype=torch.float64) * 2.5 - 2)
    qkv[:, :, 2 * D:] *= 10 ** (torch.rand(B, L, 1, generator=g, dtype=torch.float64) * 4 - 3)
    dout *= 10 ** (torch.rand(B, L, 1, generator=g, dtype=torch.float64) * 4 - 8)
    qkv = qkv.float().double().requires_grad_()
    dout = dout.float().double()
    seq = torch.randint(0, 20, (B, L), generator=g)
    seq[-1, L - 37:] = 20
    out, _ = ref_attention(qkv, seq != 20, H)
    out.backward(dout)
    qd = qkv.detach().float().view(B * L, 3 * D).to(dev)
    o, lse = K_.attention_fwd(qd, seq.to(dev), H, 0.0, 0, 0, arith=K_.GEMM_F16X2)
    dq = K_.attention_bwd(qd, seq.to(dev), o, dout.float().view(B * L, D).to(dev), lse, H, 0.0, 0, 0, arith=K_.GEMM_F16X2)
    assert torch.isfinite(o).all() and torch.isfinite(dq).all()

    def rel(a, b):
        return ((a.double().cpu() - b).norm() / b.norm()).item()
    assert rel(o.view(B, L, D), out.detach()) < 2e-6
    gr = qkv.grad
    for i, name in enumerate(("dQ", "dK", "dV")):
        assert rel(dq.view(B, L, 3 * D)[:, :, i * D:(i + 1) * D], gr[:, :, i * D:(i + 1) * D]) < 2e-6, name


@pytest.mark.parametrize("dk", [64, 32])
def test_attention_f16x2_workgroup_shapes_agree(dev, dk):
    """The launch picks the workgroup shape by the number of workgroups (attention_f16x2.hip, launch_shape): 8 wavefronts on
    256 queries / keys; when those would cover at most half of the CUs, 4 query groups x 2 halves of the key range (the
    dK/dV kernel: 4 plain wavefronts); when even 128 per workgroup would, 2 groups x 2 halves.  The halves combine partial
    soft-max states / gradient sums at the end: same dropout masks, results equal to rounding."""
    from protein_transformer_amd import kernels as K_
    B, L, H, p, seed, sid = 17, 300, 8, 0.1, 99, 7
    D = H * dk
    g = torch.Generator().manual_seed(41)
    seq = torch.randint(0, 20, (B, L), generator=g)
    seq[1, 200:] = 20
    seq[2, 33:] = 20
    qkv = torch.randn(B * L, 3 * D, generator=g).to(dev)
    dout = torch.randn(B * L, D, generator=g).to(dev)
    seq = seq.to(dev)

    def run(nb):
        n = nb * L
        o, lse = K_.attention_fwd(qkv[:n].contiguous(), seq[:nb].contiguous(), H, p, seed, sid, arith=K_.GEMM_F16X2)
        dq = K_.attention_bwd(qkv[:n].contiguous(), seq[:nb].contiguous(), o, dout[:n].contiguous(), lse, H, p, seed, sid,
                              arith=K_.GEMM_F16X2)
        return o, lse.view(nb, -1), dq
    o, lse, dq = run(B)                 # 2 * 17 * 8 = 272 workgroups of 8 wavefronts: more than the 256 CUs
    for nb in (6, 3):                   # 6: 3 * 6 * 8 = 144 workgroups of 4 x 2 (dK/dV: 4 x 1);  3: 5 * 3 * 8 = 120 of 2 x 2
        o2, lse2, dq2 = run(nb)
        n = nb * L
        assert torch.equal(o2 == 0, o[:n] == 0)       # (the dropped probabilities of a whole row can only vanish together)
        assert torch.allclose(o2, o[:n], rtol=2e-6, atol=2e-6 * o.abs().max().item()), nb
        assert torch.allclose(lse2, lse[:nb], rtol=1e-6, atol=1e-6), nb
        assert ((dq2 - dq[:n]).norm() / dq[:n].norm()).item() < 1e-6, nb
        assert torch.allclose(dq2, dq[:n], rtol=1e-4, atol=2e-6 * dq.abs().max().item()), nb
    assert torch.isfinite(dq).all() and dq.abs().max() > 0


def test_attention_f16x2_degenerate_rows(dev):
    """All-zero K / V rows (scale groups with maximum 0), a fully padded 32-key tile and huge / tiny magnitudes: finite
    results equal to the exact-f32 kernels' to rounding."""
    from protein_transformer_amd import kernels as K_
    B, L, H, dk = 2, 160, 2, 64
    D = H * dk
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(B, L, 3 * D, generator=g)
    qkv[0, 32:72, D:] = 0.0                      # zero K and V rows: whole scale groups and a whole tile of them
    qkv[1, :, 2 * D:] *= 3000.0                  # large V
    qkv[1, 5:9, 2 * D:] *= 1e-12                 # one tiny group inside
    seq = torch.randint(0, 20, (B, L), generator=g)
    seq[1, 96:] = 20                             # keys 96..159 padded: tiles 3 and 4 fully masked
    dout = torch.randn(B, L, D, generator=g) * 1e-20
    qd = qkv.view(B * L, 3 * D).to(dev)
    res = {}
    for mode in (K_.GEMM_F32, K_.GEMM_F16X2):
        o, lse = K_.attention_fwd(qd, seq.to(dev), H, 0.0, 0, 0, arith=mode)
        dq = K_.attention_bwd(qd, seq.to(dev), o, dout.view(B * L, D).to(dev), lse, H, 0.0, 0, 0, arith=mode)
        assert torch.isfinite(o).all() and torch.isfinite(dq).all() and torch.isfinite(lse).all()
        res[mode] = (o.double().cpu(), dq.double().cpu(), lse.double().cpu())
    for a, b in zip(res[K_.GEMM_F32], res[K_.GEMM_F16X2]):
        assert ((a - b).norm() / b.norm()).item() < 3e-6
    # no gradient at all (the top layer of a freshly initialised model), and none for one protein: zeros, not NaN
    for dz in (torch.zeros(B, L, D), torch.cat([torch.zeros(1, L, D), torch.randn(1, L, D, generator=g)])):
        o, lse = K_.attention_fwd(qd, seq.to(dev), H, 0.1, 3, 1, arith=K_.GEMM_F16X2)
        dq = K_.attention_bwd(qd, seq.to(dev), o, dz.view(B * L, D).to(dev), lse, H, 0.1, 3, 1, arith=K_.GEMM_F16X2)
        assert torch.isfinite(dq).all()
        assert (dq.view(B, L, 3 * D)[0] == 0).all()
    vz = qkv.clone()
    vz[:, :, 2 * D:] = 0.0                       # V == 0 everywhere: the forward pass never sees a non-zero scale group
    o, lse = K_.attention_fwd(vz.view(B * L, 3 * D).to(dev), seq.to(dev), H, 0.0, 0, 0, arith=K_.GEMM_F16X2)
    assert torch.isfinite(o).all() and (o == 0).all()


# ------------------------------------------------------------------------------------------------ optimizer
@pytest.mark.parametrize("n", [1000003, 4096])
def test_clip_and_sgd_adam(dev, n):
    from protein_transformer_amd import kernels as K_
    w0, g = rnd((n,), 30), rnd((n,), 31, 0.01 if n > 5000 else 5.0)
    for max_norm in (1.0, 1e9):
        ref = w0.clone().requires_grad_()
        ref.grad = g.clone()
        opt = torch.optim.SGD([ref], lr=1e-2, weight_decay=0.01)
        total = torch.nn.utils.clip_grad_norm_([ref], max_norm)
        opt.step()
        w = w0.clone().to(dev)
        sq = torch.zeros(1, device=dev)
        K_.grad_sqnorm(g.to(dev), sq)
        assert sq.sqrt().item() == pytest.approx(float(total), rel=1e-5)
        K_.sgd_step(w, g.to(dev), sq, max_norm, 1e-2, 0.01)
        assert_close(w, ref, 1e-6, 1e-7, "sgd")
    # Adam's update lr * m / (sqrt(v) + eps) is scale free in g' = clip*g + wd*w, so elements whose g' cancels to
    # ~1 ulp are ill-conditioned (a 1e-6 relative difference in the fp32 norm flips them).  Use same-signed w and g
    # so that g' never cancels and the comparison tests the kernel, not the conditioning.
    w0, g = w0.abs() + 0.1, g.abs() + 1e-4
    ref = w0.clone().requires_grad_()
    opt = torch.optim.Adam([ref], betas=(0.9, 0.98), eps=1e-9, lr=1e-3, weight_decay=0.01)
    w, m, v = w0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    sq = torch.zeros(1, device=dev)
    for step in (1, 2, 3):
        gi = g * step
        ref.grad = gi.clone()
        torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        K_.grad_sqnorm(gi.to(dev), sq)
        K_.adam_step(w, gi.to(dev), m, v, sq, 1.0, 1e-3, 0.9, 0.98, 1e-9, 0.01, step)
    assert_close(w, ref, 1e-5, 1e-6, "adam")
    assert_close(m, opt.state[ref]["exp_avg"], 1e-4, 1e-9, "adam m")
    assert_close(v, opt.state[ref]["exp_avg_sq"], 1e-4, 1e-12, "adam v")
