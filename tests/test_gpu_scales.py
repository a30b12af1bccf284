"""MI355X tests of the f16x2 bookkeeping that replaces the per-product pass over the operands (csrc/scales.hip, the
row-scale output of ptamd_layernorm_fwd, ptamd_layernorm_bwd_dropout, ptamd_gemm_args.a_scale / b_scale):

  * weight scales / statistics against numpy;
  * the weight-derived BOUNDS really bound the attention output, the FFN hidden layer and its gradient on a model with
    trained-looking (non-trivial) LayerNorm parameters, and are within a few binades of the true maxima;
  * LayerNorm backward fused with the dropout backward == the two separate kernels, bit for bit, and its row scales are the
    exact ones;
  * a product with caller-provided exact scales == the product that finds them itself, bit for bit; with a bound instead
    of the maximum it stays inside the f16x2 error model.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def scale_of(amax):
    """numpy restatement of pt_row_scale_bits: the power of two that takes amax into [2^14, 2^15)."""
    amax = np.asarray(amax, np.float32)
    e = (amax.view(np.uint32) >> 23).astype(np.int64)
    return np.ldexp(1.0, np.minimum(268 - e, 254) - 127)


def as_float(bits):
    return bits.cpu().numpy().view(np.float32).astype(np.float64)


def test_weight_scales_vs_numpy(dev):
    from protein_transformer_amd import kernels as K
    g = torch.Generator().manual_seed(1)
    flat = (torch.randn(2048 * 512 + 4096, generator=g) * torch.exp(torch.randn(2048 * 512 + 4096, generator=g))).to(dev)
    w1 = flat[:2048 * 512].view(2048, 512)
    sub = w1[700:1200]                                   # a sub-matrix (rows 700..1199), like W_v inside W_qkv
    vec = flat[2048 * 512:2048 * 512 + 1000]
    rs, cs = torch.zeros(2048, dtype=torch.int32, device=dev), torch.zeros(512, dtype=torch.int32, device=dev)
    st = torch.full((3, 4), -1.0, device=dev)
    K.weight_scales([dict(w=w1, row_scale=rs, col_scale=cs, stats=st[0]), dict(w=sub, stats=st[1]), dict(w=vec, stats=st[2])])
    w = w1.cpu().numpy()
    assert np.array_equal(as_float(rs), scale_of(np.abs(w).max(1)))
    assert np.array_equal(as_float(cs), scale_of(np.abs(w).max(0)))
    s = st.cpu().numpy().astype(np.float64)
    w64 = w.astype(np.float64)
    assert s[0, 0] == pytest.approx(np.sqrt((w64 ** 2).sum(1)).max(), rel=1e-5)
    assert s[0, 1] == pytest.approx(np.sqrt((w64 ** 2).sum(0)).max(), rel=1e-5)
    assert s[0, 2] == np.abs(w).max() and s[0, 3] == 0
    assert s[1, 0] == pytest.approx(np.sqrt((w64[700:1200] ** 2).sum(1)).max(), rel=1e-5)
    v = vec.cpu().numpy().astype(np.float64)
    assert s[2, 0] == pytest.approx(np.sqrt((v ** 2).sum()), rel=1e-5) and s[2, 2] == np.abs(v).max()
    top = np.abs(w).max(1) * as_float(rs)
    assert np.all((top >= 2.0 ** 14) & (top < 2.0 ** 15))


def test_layernorm_fwd_row_scale(dev):
    from protein_transformer_amd import kernels as K
    x = torch.randn(1000, 512, device=dev) * 3 + 1
    gam, bet = torch.rand(512, device=dev) + 0.5, torch.randn(512, device=dev) * 0.1
    s = torch.zeros(1000, dtype=torch.int32, device=dev)
    y, _, _ = K.layernorm_fwd(x, gam, bet, row_scale=s)
    y0, _, _ = K.layernorm_fwd(x, gam, bet)
    assert torch.equal(y, y0)
    assert np.array_equal(as_float(s), scale_of(y.abs().max(1).values.cpu().numpy()))


# (16384 tokens: one wavefront per 8-row generator group; 8000: two; the smaller ones: four - elementwise.hip, HALVES)
@pytest.mark.parametrize("T,D,p", [(16384, 512, 0.1), (8000, 512, 0.1), (1000, 256, 0.3), (77, 64, 0.1), (640, 1024, 0.0), (333, 512, 0.0)])
def test_layernorm_bwd_dropout_equals_separate_kernels(dev, T, D, p):
    from protein_transformer_amd import kernels as K
    g = torch.Generator().manual_seed(T + D)
    x, dy, dres = (torch.randn(T, D, generator=g).to(dev) for _ in range(3))
    dy = dy * torch.exp(torch.randn(T, 1, generator=g) * 2).to(dev)                      # per-token gradient magnitudes
    gam = (torch.rand(D, generator=g) + 0.5).to(dev)
    _, mean, rstd = K.layernorm_fwd(x, gam, torch.zeros(D, device=dev))
    dg0, db0 = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    dx0 = K.layernorm_bwd(dy, x, gam, mean, rstd, dg0, db0, dres=dres)
    dr0 = K.dropout_bwd(dx0, p, 1234, 13) if p > 0 else dx0
    dg1, db1 = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    rs, bs = torch.zeros(T, dtype=torch.int32, device=dev), torch.zeros(T, dtype=torch.int32, device=dev)
    factor = torch.tensor([3.5], device=dev)
    dx1, dr1 = K.layernorm_bwd_dropout(dy, x, gam, mean, rstd, dg1, db1, dres, p, 1234, 13, row_scale=rs, bound_factor=factor,
                                       bound_scale=bs)
    # the two kernels evaluate the same formulas with different fma contractions (the fused one loads unconditionally and
    # software-pipelines its rows): equal to rounding relative to the size of the row, identical dropout pattern
    tol = 2e-6 * dx0.abs().max(1, keepdim=True).values
    assert bool(((dx1 - dx0).abs() <= tol).all()) and bool(((dr1 - dr0).abs() <= tol / (1 - p)).all())
    assert bool((((dr1 == 0) == (dr0 == 0)) | (dx0.abs() <= tol)).all())       # (a dx that cancels to exactly 0 in one of them)
    # (rows are dealt to the wavefronts in another order: the partial sums differ in their rounding)
    assert float((dg1 - dg0).abs().max()) <= 2e-5 * float(dg0.abs().max()) and float((db1 - db0).abs().max()) <= 2e-5 * float(db0.abs().max())
    assert np.array_equal(as_float(rs), scale_of(dr1.abs().max(1).values.cpu().numpy()))     # exact maxima of what was written
    nrm = np.sqrt((dr1.cpu().double().numpy() ** 2).sum(1)) * 3.5
    got = as_float(bs)
    want = scale_of(nrm.astype(np.float32))
    assert np.all((got == want) | (got == want * 2) | (got * 2 == want))                 # fp32 rounding of the norm at a binade edge
    assert np.all(nrm * got < 2.0 ** 15 * (1 + 1e-6))


def test_gemm_with_caller_scales(dev):
    from protein_transformer_amd import kernels as K
    g = torch.Generator().manual_seed(5)
    M, N, Kd = 1024, 384, 512
    a = (torch.randn(M, Kd, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(dev)
    w = (torch.randn(N, Kd, generator=g) * 0.05).to(dev)
    bias = torch.randn(N, generator=g).to(dev)

    def run(**kw):
        c = torch.empty(M, N, device=dev)
        K.gemm(a, w, c, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, bias=bias, arith=K.GEMM_F16X2, **kw)
        return c
    ref = run()
    sa = torch.zeros(M, dtype=torch.int32, device=dev)
    K.weight_scales([dict(w=a, row_scale=sa)])
    rs = torch.zeros(N, dtype=torch.int32, device=dev)
    K.weight_scales([dict(w=w, row_scale=rs)])
    assert torch.equal(run(a_scale=sa, b_scale=rs), ref)                                 # exact scales: the same arithmetic
    assert torch.equal(run(a_scale=sa), ref) and torch.equal(run(b_scale=rs), ref)       # one operand provided, one found
    # dX layout: B k-major, scale per output column
    dy = (torch.randn(M, N, generator=g)).to(dev)
    cs = torch.zeros(Kd, dtype=torch.int32, device=dev)
    K.weight_scales([dict(w=w, col_scale=cs)])
    dx0 = K.linear_bwd_input(dy, w, arith=K.GEMM_F16X2)
    assert torch.equal(K.linear_bwd_input(dy, w, arith=K.GEMM_F16X2, b_scale=cs), dx0)
    # a bound 2^5 above the largest row maximum, one scale for all rows: inside the norm-wise error model
    bound = torch.tensor([float(a.abs().max()) * 32], device=dev)
    ub = torch.zeros(1, dtype=torch.int32, device=dev)
    K.weight_scales([dict(w=bound, row_scale=ub)])
    got = run(a_scale=ub, a_scale_stride=0, b_scale=rs)
    exact = a.double() @ w.double().t() + bias.double()
    allowed = 2.0 ** -20 * (a.double().abs() @ w.double().abs().t()) + 2.0 ** -36 * Kd * float(bound) * w.abs().max(1).values.double()[None, :] + 1e-6 * exact.abs()
    assert bool(((got.double() - exact).abs() <= allowed).all())


def test_weight_derived_bounds_hold_and_are_tight(dev):
    """The scales the model computes from its weights alone (attention output, FFN hidden layer, FFN hidden gradient):
    never below what the tensors need (no f16 overflow), and within 2^7 of the exact row scale."""
    from protein_transformer_amd import kernels as K
    from protein_transformer_amd.models.encoder_only import EncoderOnlyTransformer
    from protein_transformer_amd.protein.Sequence import VOCAB
    torch.manual_seed(2)
    m = EncoderOnlyTransformer(2, 8, 512, 2048, 128, VOCAB, np.zeros(24) + 0.3, True, dropout=0.1).to(dev).train()
    with torch.no_grad():                                  # LayerNorm parameters away from (1, 0), biases away from 0
        for n, q in m.named_parameters():
            if "norm.weight" in n:
                q.uniform_(0.5, 2.0)
            elif "norm.bias" in n:
                q.normal_(0, 0.3)
            elif n.endswith("bias"):
                q.normal_(0, 0.5)
    flat, _ = m.flat_parameters()
    p, pa = 0.1, 0.1
    layers = m._step_scales(flat, K.GEMM_AUTO, p, pa)
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    x = torch.randn(4 * 128, 512, device=dev) * 5
    for i, L in enumerate(layers):
        b = f"encoder.enc_layers.{i}."
        ln = lambda t, j: torch.nn.functional.layer_norm(t, (512,), sd[b + f"sublayer_connections.{j}.norm.weight"],      # noqa: E731
                                                         sd[b + f"sublayer_connections.{j}.norm.bias"], 1e-5)
        h1 = ln(x, 0)
        v = h1 @ sd[b + "self_attn.wv.weight"].t() + sd[b + "self_attn.wv.bias"]
        att_max = float(v.abs().max()) / (1 - pa)          # every row of the attention output is below this
        s_att = float(as_float(L["att_scale"])[0])
        assert att_max * s_att < 2.0 ** 15 and att_max * s_att > 2.0 ** 7, (i, att_max * s_att)
        h2 = ln(x * 0.3 + 1, 1)
        f1 = torch.relu(h2 @ sd[b + "pwff.layer1.weight"].t() + sd[b + "pwff.layer1.bias"]) / (1 - p)
        s_f1 = float(as_float(L["f1_scale"])[0])
        assert float(f1.max()) * s_f1 < 2.0 ** 15 and float(f1.max()) * s_f1 > 2.0 ** 8
        dy = torch.randn(512, 512, device=dev) * torch.exp(torch.randn(512, 1, device=dev) * 3)
        dz = (dy @ sd[b + "pwff.layer2.weight"]) / (1 - p)
        fac = float(L["dz1_factor"][0])
        ratio = (dy.norm(dim=1) * fac) / dz.abs().max(1).values
        assert float(ratio.min()) >= 1.0 and float(ratio.max()) < 2.0 ** 7


def test_weight_gradient_product_with_uniform_scales(dev):
    """dW = dy^T x in f16x2 arithmetic with ONE scale per operand (the scale of its largest row - four copies, stride 0):
    per-token gradients spanning four decades, activations with a bound 2^4 above their maximum.  Norm-wise fp32-grade:
    the error against fp64 is measured in units of the exact product's own largest entries, next to the bf16x3 result."""
    from protein_transformer_amd import kernels as K
    g = torch.Generator().manual_seed(11)
    T, N, Kd = 16384, 512, 2048
    dy = (torch.randn(T, N, generator=g) * torch.exp(torch.randn(T, 1, generator=g) * 2.3)).to(dev)   # 1e-4 .. 1e4 per token
    x = torch.relu(torch.randn(T, Kd, generator=g)).to(dev)
    uni = torch.zeros(2, 4, dtype=torch.int32, device=dev)
    st = torch.zeros(2, 4, device=dev)
    K.weight_scales([dict(w=dy, stats=st[0], rows_only=True), dict(w=x, stats=st[1], rows_only=True)])
    K.bound_scales([dict(w=st[0], w_index=2, out_scale=uni[0]), dict(w=st[1], w_index=2, post_scale=16.0, out_scale=uni[1])])
    assert float(as_float(uni[0])[0]) * float(dy.abs().max()) < 2.0 ** 15 and len(set(uni[0].tolist())) == 1
    dw_u, dw_b = torch.zeros(N, Kd, device=dev), torch.zeros(N, Kd, device=dev)
    db_u, db_b = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    K.linear_bwd_weight(dy, x, dw_u, db_u, dy_scale=uni[0], x_scale=uni[1])
    K.linear_bwd_weight(dy, x, dw_b, db_b, arith=K.GEMM_BF16X3)
    ref = dy.double().t() @ x.double()
    unit = float(ref.abs().max())
    e_u, e_b = float((dw_u.double() - ref).abs().max()) / unit, float((dw_b.double() - ref).abs().max()) / unit
    print("dW error / max|dW|: f16x2 with uniform scales", e_u, " bf16x3", e_b)
    assert e_b < 1e-6 and e_u < 2e-6
    rel = float((dw_u.double() - ref).norm() / ref.norm())
    assert rel < 1e-6
    assert torch.allclose(db_u, db_b, rtol=1e-5, atol=1e-3 * float(db_b.abs().max()))      # the fused bias gradient is fp32 either way


@pytest.mark.parametrize("T,shapes,split", [(16384, [(512, 2048), (2048, 512), (512, 512), (1536, 512)], 8),
                                            (4096, [(256, 1024), (1024, 256), (768, 256)], 3),
                                            (2100, [(512, 2048), (36, 260)], 2)])
def test_weight_gradient_products_as_a_group(dev, T, shapes, split):
    """ptamd_gemm_group: the weight-gradient products of a layer in one launch (+ one reduce launch) give, bit for bit, what
    the separate ptamd_gemm calls with the same K split give - accumulated into non-zero dW / dbias, ragged tiles and a
    token count that is not a multiple of the stage included; members it does not take are refused before any launch."""
    import ctypes as C
    from protein_transformer_amd import kernels as K
    from protein_transformer_amd._lib import GemmArgs, lib
    g = torch.Generator().manual_seed(7)
    jobs, sep = [], []
    for N, Kd in shapes:
        dy = (torch.randn(T, N, generator=g) * torch.exp(torch.randn(T, 1, generator=g))).to(dev)
        x = torch.randn(T, Kd, generator=g).to(dev)
        uni = torch.zeros(2, 4, dtype=torch.int32, device=dev)
        st = torch.zeros(2, 4, device=dev)
        K.weight_scales([dict(w=dy, stats=st[0], rows_only=True), dict(w=x, stats=st[1], rows_only=True)])
        K.bound_scales([dict(w=st[0], w_index=2, out_scale=uni[0]), dict(w=st[1], w_index=2, post_scale=4.0, out_scale=uni[1])])
        dw0, db0 = torch.randn(N, Kd, generator=g).to(dev), torch.randn(N, generator=g).to(dev)
        jobs.append((dy, x, dw0.clone(), db0.clone(), uni[0], uni[1]))
        sep.append((dy, x, dw0.clone(), db0.clone(), uni[0], uni[1]))
    K.linear_bwd_weight_group(jobs, split)
    for dy, x, dw, db, sy, sx in sep:
        N, Kd = dw.shape
        K.gemm(dy, x, dw, M=N, N=Kd, K=T, lda=N, ldb=Kd, ldc=Kd, a_kmajor=True, b_kmajor=True, flags=K.EPI_ACCUM, split_k=split,
               colsum=db, arith=K.GEMM_F16X2, a_scale=sy, a_scale_stride=0, b_scale=sx, b_scale_stride=0)
    torch.cuda.synchronize()
    for (_, _, dw_g, db_g, _, _), (dy, x, dw_s, db_s, _, _) in zip(jobs, sep):
        assert torch.equal(dw_g, dw_s) and torch.equal(db_g, db_s)
    dy, x, dw, db, sy, sx = jobs[0]
    # refused: bf16x3 member, a member without slabs, too many members
    def args(**kw):
        N, Kd = dw.shape
        ws = torch.empty(lib().ptamd_gemm_workspace_bytes(N, Kd, split), dtype=torch.uint8, device=dev)
        base = dict(M=N, N=Kd, K=T, A=dy.data_ptr(), lda=N, a_kmajor=1, B=x.data_ptr(), ldb=Kd, b_kmajor=1, C=dw.data_ptr(), ldc=Kd,
                    bias=None, residual=None, ldr=0, flags=K.EPI_ACCUM, dropout_p=0.0, seed=0, stream_id=0, split_k=split,
                    workspace=ws.data_ptr(), workspace_bytes=ws.numel(), colsum=None, gate_scale=0.0, arith=K.GEMM_F16X2,
                    reserved_cus=0, a_scale=sy.data_ptr(), a_scale_stride=0, b_scale=sx.data_ptr(), b_scale_stride=0)
        base.update(kw)
        return GemmArgs(**base), ws
    before = dw.clone()
    for bad in (dict(arith=K.GEMM_BF16X3), dict(split_k=1), dict(flags=0), dict(a_scale=None), dict(a_kmajor=0, lda=T)):
        a, _ws = args(**bad)
        arr = (GemmArgs * 1)(a)
        assert lib().ptamd_gemm_group(arr, 1, K.stream()) == -1, bad
    a, _ws = args()
    assert lib().ptamd_gemm_group((GemmArgs * 5)(a, a, a, a, a), 5, K.stream()) == -1
    assert lib().ptamd_gemm_group(None, 1, K.stream()) == -1
    torch.cuda.synchronize()
    assert torch.equal(dw, before)
    assert K.pick_group_split(64, 16384) == 4 and K.pick_group_split(32, 16384) == 8 and K.pick_group_split(96, 1024) == 0


@pytest.mark.parametrize("B,L,H,dk", [(3, 300, 4, 64), (2, 130, 8, 32)])
def test_attention_bwd_row_scales(dev, B, L, H, dk):
    """ptamd_attention_bwd (f16x2 arithmetic) leaves the f16x2 row scales of dqkv and the smallest of them behind: exactly
    those of a pass over the dqkv it wrote; other arithmetics refuse the request."""
    from protein_transformer_amd import kernels as K
    g = torch.Generator().manual_seed(3)
    D = H * dk
    qkv = torch.randn(B * L, 3 * D, generator=g).to(dev)
    dout = (torch.randn(B * L, D, generator=g) * torch.exp(2 * torch.randn(B * L, 1, generator=g))).to(dev) * 1e-3
    seq = torch.randint(0, 20, (B, L), generator=g)
    seq[0, L - 40:] = 20
    seq = seq.to(dev)
    o, lse = K.attention_fwd(qkv, seq, H, 0.1, 5, 2, arith=K.GEMM_F16X2)
    rs = torch.full((B * L,), 0x7F000000, dtype=torch.int32, device=dev)
    mn = torch.full((4,), 0x7F000000, dtype=torch.int32, device=dev)
    dq = K.attention_bwd(qkv, seq, o, dout, lse, H, 0.1, 5, 2, arith=K.GEMM_F16X2, row_scale=rs, row_scale_min=mn)
    dq0 = K.attention_bwd(qkv, seq, o, dout, lse, H, 0.1, 5, 2, arith=K.GEMM_F16X2)
    assert torch.equal(dq, dq0)                                                  # the by-product does not touch the result
    want = scale_of(dq.abs().amax(dim=1).cpu().numpy())
    assert np.array_equal(as_float(rs), want)
    assert np.array_equal(as_float(mn), np.full(4, want.min()))
    assert K.attention_row_scales_available(dk, K.GEMM_AUTO) and not K.attention_row_scales_available(dk, K.GEMM_BF16X3)
    with pytest.raises(RuntimeError):
        K.attention_bwd(qkv, seq, o, dout, lse, H, 0.1, 5, 2, arith=K.GEMM_BF16X3, row_scale=rs, row_scale_min=mn)


def _prep_model(dev, nl=2, dm=512, dff=2048, seed=3, nprot=8):
    from protein_transformer_amd import synthetic
    from protein_transformer_amd.models.encoder_only import EncoderOnlyTransformer
    from protein_transformer_amd.protein.Sequence import VOCAB
    from protein_transformer_amd.protein.Structure import nerf_forward
    build = lambda ang, seq: nerf_forward(ang.to(dev), seq.to(dev))[0]  # noqa: E731
    batch = synthetic.make_batch([512] * nprot, L_pad=512, seed=seed, build_coords=build)
    torch.manual_seed(seed)
    m = EncoderOnlyTransformer(nl, 8, dm, dff, 512, VOCAB, synthetic.angle_means(batch["true_ang"]), True, dropout=0.1).to(dev).train()
    with torch.no_grad():
        P = dict(m.named_parameters())
        P["output_projection.weight"].normal_(0, 0.02)
        for n, p in P.items():
            if "norm.weight" in n:
                p.add_(0.1 * torch.randn_like(p))
            elif "norm.bias" in n or n.endswith(".bias"):
                p.add_(0.05 * torch.randn_like(p))
    return m, tuple(batch[k].to(dev) for k in ("seq", "true_ang", "true_crd"))


def _snapshot(layers):
    out = {}
    for i, L in enumerate(layers):
        for k in ("rs_qkv", "cs_qkv", "rs_o", "cs_o", "rs_1", "cs_1", "rs_2", "cs_2", "att_scale", "f1_scale", "h1_scale", "h2_scale"):
            out[(i, k)] = L[k].clone()
        out[(i, "dz1_factor")] = L["dz1_factor"].clone()
        for k in ("hp_1", "hp_qkv", "hp_2t"):
            out[(i, k)] = L[k].planes.clone()
    return out


def test_weights_prep_matches_separate_launches(dev):
    """csrc/wprep.hip (ONE pass over the weights, two launches) against the launches it replaces (ptamd_weight_scales,
    ptamd_bound_scales, ptamd_hp_split_rows, ptamd_hp_split_cols) on a model's weights: every row / column / bound scale
    and every plane byte identical - also the one statistic whose summation order differs (the largest column norm of W_2: both
    kernels sum its squares in fp64, so the fp32 results agree whatever the order).  Also with the vector segments at odd offsets (b_v inside the QKV bias) and twice in
    a row (the two copies of the atomicMax targets alternate)."""
    from protein_transformer_amd import kernels as K
    m, _ = _prep_model(dev)
    flat, _ = m.flat_parameters()
    m.__dict__["_fwd_grad"] = True
    m.weights_prep = False
    ref = _snapshot(m._step_scales(flat, K.GEMM_AUTO, 0.1, 0.1, hp=True))
    for L in m._step_scales(flat, K.GEMM_AUTO, 0.1, 0.1, hp=False):      # scribble over everything the prep pass has to write
        for k in ("rs_qkv", "cs_qkv", "rs_o", "cs_o", "rs_1", "cs_1", "rs_2", "cs_2", "att_scale", "f1_scale", "h1_scale", "h2_scale"):
            L[k].fill_(-7)
        L["dz1_factor"].fill_(-1.0)
        for k in ("hp_1", "hp_qkv", "hp_2t"):
            L[k].planes.fill_(0x5A)
    m.weights_prep = True
    for rep in range(3):
        m._forget_prepared_weights()
        got = _snapshot(m._step_scales(flat, K.GEMM_AUTO, 0.1, 0.1, hp=True))
        torch.cuda.synchronize()
        for key, want in ref.items():
            assert torch.equal(got[key], want), (rep, key, int((got[key] != want).sum()))
    assert m.__dict__["_prep_launches"] == 3
    # nothing has touched the weights: the next pass launches nothing; an in-place torch op on a parameter is noticed
    m._step_scales(flat, K.GEMM_AUTO, 0.1, 0.1, hp=True)
    assert m.__dict__["_prep_launches"] == 3
    with torch.no_grad():
        dict(m.named_parameters())["encoder.enc_layers.1.pwff.layer2.weight"][5, 7] = 1000.0
    L = m._step_scales(flat, K.GEMM_AUTO, 0.1, 0.1, hp=True)
    assert m.__dict__["_prep_launches"] == 4
    assert as_float(L[1]["cs_2"])[7] == scale_of(np.float32(1000.0)) and as_float(L[1]["rs_2"])[5] == scale_of(np.float32(1000.0))


@pytest.mark.parametrize("optimizer", ["sgd", "adam"])
def test_optimizer_step_prepares_the_next_pass(dev, optimizer):
    """ptamd_sgd_step_prep / ptamd_adam_step_prep: the optimizer step that also leaves the scales / bounds / planes of the NEW
    weights behind.  Two identical models, one with the fused step and one with the separate launches of rounds 2-4, four
    training steps each: parameters and gradients bit-identical after every step (same update expression, same scales), the
    fused one launched its preparation pass ONCE (the first forward pass; afterwards the optimizer step did it), and what the
    step left behind is what a fresh pass over the updated weights computes."""
    import types
    from protein_transformer_amd import kernels as K
    from protein_transformer_amd.optim import FusedAdam, FusedSGD
    from protein_transformer_amd.train import train_step
    args = types.SimpleNamespace(loss="drmsd", combined_drmsd_weight=0.5, backbone_loss=False, clip=1.0)
    runs = {}
    for prep in (True, False):
        m, data = _prep_model(dev, seed=5)
        m.weights_prep = prep
        opt = (FusedAdam(m, lr=1e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=10e-3) if optimizer == "adam"
               else FusedSGD(m, lr=1e-2, weight_decay=10e-3))
        snaps = []
        for _ in range(4):
            train_step(m, opt, args, *data)
            snaps.append((m.flat_parameters()[0].clone(), m.flat_parameters()[1].clone()))
        runs[prep] = (m, snaps)
    for (wa, ga), (wb, gb) in zip(runs[True][1], runs[False][1]):
        assert torch.equal(ga, gb) and torch.equal(wa, wb)
    m = runs[True][0]
    assert m.__dict__.get("_prep_launches", 0) == 1
    assert runs[False][0].__dict__.get("_prep_launches", 0) == 0
    flat, _ = m.flat_parameters()
    left = _snapshot(m.__dict__["_train_cache"][0]["layers"])
    m._forget_prepared_weights()
    m.__dict__["_fwd_grad"] = True
    again = _snapshot(m._step_scales(flat, K.GEMM_AUTO, m.dropout, m.attn_dropout, hp=True))
    for key, want in again.items():
        assert torch.equal(left[key], want), key


@pytest.mark.parametrize("optimizer", ["sgd", "adam"])
def test_eval_between_training_steps_sees_the_current_weights(dev, optimizer):
    """Round-5 advisor finding (high): the fused optimizer kernels write the flat buffer through a raw pointer, so torch's
    version counters - all that `_weights_stamp` looked at - never moved during training, and the EVALUATION pass's scale cache
    (dropout 0: another cache entry than the training pass's) kept the scales / bounds / planes of the weights of the first
    validation.  train -> eval -> train -> eval with the fused step against the same sequence with `weights_prep = False`
    (every forward pass derives everything from the weights as they are): evaluation outputs bit-identical at every stage,
    and the second evaluation differs from the first (the weights moved)."""
    import types
    from protein_transformer_amd.optim import FusedAdam, FusedSGD
    from protein_transformer_amd.train import train_step
    args = types.SimpleNamespace(loss="drmsd", combined_drmsd_weight=0.5, backbone_loss=False, clip=1.0)
    outs = {}
    for prep in (True, False):
        m, data = _prep_model(dev, seed=11)
        m.weights_prep = prep
        opt = (FusedAdam(m, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=10e-3) if optimizer == "adam"
               else FusedSGD(m, lr=5e-2, weight_decay=10e-3))
        got = []
        for _ in range(3):
            train_step(m, opt, args, *data)
            train_step(m, opt, args, *data)
            m.eval()
            with torch.no_grad():
                got.append(m(data[0]).clone())
            m.train()
        outs[prep] = got
        if prep:
            # the evaluation cache prepared the weights again after every pair of steps (its stamp moved with the steps);
            # the training cache only for the model's first pass
            assert m.__dict__.get("_prep_launches", 0) == 1 + 3
    for a, b in zip(outs[True], outs[False]):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    assert not torch.equal(outs[True][0], outs[True][1]) and not torch.equal(outs[True][1], outs[True][2])


def test_set_dropout_and_plain_steps_invalidate_prepared_weights(dev):
    """... the same staleness behind `set_dropout` (another cache key) and behind the PLAIN optimizer kernel (a step taken
    while `prepared_step` has nothing to offer - here: `weights_prep` switched off for one step): the pass after it must
    not find a cache 'fresh' that was prepared on older weights."""
    import types
    from protein_transformer_amd.optim import FusedSGD
    from protein_transformer_amd.train import train_step
    args = types.SimpleNamespace(loss="drmsd", combined_drmsd_weight=0.5, backbone_loss=False, clip=1.0)
    res = {}
    for prep in (True, False):
        m, data = _prep_model(dev, seed=12)
        m.weights_prep = prep
        opt = FusedSGD(m, lr=5e-2, weight_decay=10e-3)
        m.set_dropout(0.0)
        train_step(m, opt, args, *data)          # cache (0, 0): prepared by the pass, then by the step
        m.set_dropout(0.1, 0.1)
        train_step(m, opt, args, *data)          # cache (0.1, 0.1): first use
        m.set_dropout(0.0)
        train_step(m, opt, args, *data)          # cache (0, 0) again: what it holds is one step old
        m.weights_prep = False
        train_step(m, opt, args, *data)          # the plain kernel writes the weights
        m.weights_prep = prep
        train_step(m, opt, args, *data)          # ... and the cache that was fresh two steps ago is not
        res[prep] = (m.flat_parameters()[0].clone(), m.flat_parameters()[1].clone())
    assert torch.isfinite(res[True][0]).all()
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])


@pytest.mark.parametrize("dm,dff,nh", [(768, 2048, 12), (512, 1000, 8), (256, 768, 8)])
def test_widths_the_one_pass_preparation_cannot_panel(dev, dm, dff, nh):
    """Round-5 advisor finding (medium): csrc/wprep.hip walks matrices wider than 512 columns as 512-column panels; a width
    above 512 that is not a multiple of 512 (`-dm 768`, d_ff 1000) raised ValueError in the first forward pass although
    `weights_prep = True` is the default.  Such models take the separate launches (`prep` is None) and train."""
    import types
    from protein_transformer_amd import synthetic
    from protein_transformer_amd.models.encoder_only import EncoderOnlyTransformer
    from protein_transformer_amd.optim import FusedSGD
    from protein_transformer_amd.protein.Sequence import VOCAB
    from protein_transformer_amd.protein.Structure import nerf_forward
    from protein_transformer_amd.train import train_step
    build = lambda ang, seq: nerf_forward(ang.to(dev), seq.to(dev))[0]  # noqa: E731
    batch = synthetic.make_batch([512] * 8, L_pad=512, seed=4, build_coords=build)
    data = tuple(batch[k].to(dev) for k in ("seq", "true_ang", "true_crd"))
    torch.manual_seed(4)
    args = types.SimpleNamespace(loss="drmsd", combined_drmsd_weight=0.5, backbone_loss=False, clip=1.0)
    flats = []
    for prep in (True, False):
        torch.manual_seed(4)
        m = EncoderOnlyTransformer(2, nh, dm, dff, 512, VOCAB, synthetic.angle_means(batch["true_ang"]), True, dropout=0.1).to(dev).train()
        m.weights_prep = prep
        opt = FusedSGD(m, lr=1e-2, weight_decay=10e-3)
        for _ in range(2):
            losses = train_step(m, opt, args, *data)
        assert np.isfinite(float(losses["drmsd-full"]))
        caches = m.__dict__.get("_scale_caches", {})
        if prep and caches:
            narrow = all(c <= 512 or c % 512 == 0 for c in (dm, dff))
            assert all((c["prep"] is not None) == narrow for c in caches.values())
        flats.append(m.flat_parameters()[0].clone())
    assert torch.isfinite(flats[0]).all() and torch.equal(flats[0], flats[1])


@pytest.mark.parametrize("optimizer", ["sgd", "adam"])
def test_step_zeroes_the_gradient_for_the_next_step(dev, optimizer):
    """`zero_grad_in_step` (optim.py; ptamd_*_step*'s `zero_grad` argument): the optimizer step writes the zeros the next
    step's `optimizer.zero_grad()` would write, and that call then finds nothing to do.  Same trajectory bit for bit as with
    the fill; the buffer IS zero behind the step; a backward pass (raw-pointer writes) or a torch op on a `p.grad` between the
    step and the next `zero_grad()` makes that call fill again."""
    import types
    from protein_transformer_amd.optim import FusedAdam, FusedSGD
    from protein_transformer_amd.train import get_losses, train_step
    args = types.SimpleNamespace(loss="drmsd", combined_drmsd_weight=0.5, backbone_loss=False, clip=1.0)
    runs = {}
    for fused in (True, False):
        m, data = _prep_model(dev, seed=21)
        opt = (FusedAdam(m, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=10e-3) if optimizer == "adam"
               else FusedSGD(m, lr=1e-2, weight_decay=10e-3))
        assert opt.zero_grad_in_step is False                # torch's semantics unless the training loop asks
        opt.zero_grad_in_step = fused
        snaps = []
        for k in range(4):
            train_step(m, opt, args, *data)
            _, g = m.flat_parameters()
            if fused:
                assert not bool(g.any())
            else:
                assert bool(g.any())
            if k == 1:
                # a backward pass outside the loop (no zero_grad in front of it, like a gradient probe): the next step's
                # zero_grad must clear what it left
                get_losses(args, m(data[0], data[1]), data[1], data[2], data[0])
                assert bool(g.any())
            if k == 2:
                next(m.parameters()).grad.add_(1.0)          # ... and a torch op on a gradient view
            snaps.append(m.flat_parameters()[0].clone())
        runs[fused] = snaps
    for a, b in zip(runs[True], runs[False]):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_fill_u32(dev):
    from protein_transformer_amd import kernels as K
    a = torch.zeros(6, 16, dtype=torch.int32, device=dev)
    b = torch.zeros(3, 1001, dtype=torch.int32, device=dev)
    c = torch.zeros(7, dtype=torch.float32, device=dev)[1:]      # unaligned start
    K.fill_u32([(a, 0x7F000000), (b, 5), (c, 0x3F800000)])
    torch.cuda.synchronize()
    assert bool((a == 0x7F000000).all()) and bool((b == 5).all()) and bool((c == 1.0).all())


@pytest.mark.parametrize("T,D,N,want", [(2048, 512, 2048, 4), (2048, 512, 1536, 3), (4096, 512, 2048, 2), (1900, 256, 2048, 4),
                                        (16384, 512, 2048, 0)])
def test_unreduced_k_slices_into_the_layernorm_backward(dev, T, D, N, want):
    """Round 6 (PTAMD_EPI_SLABS, ptamd_layernorm_bwd_dropout: dy_slabs): at few tokens the dX product in front of a fused
    LayerNorm backward leaves its K slices unreduced and that kernel adds them in slab order as it reads the rows - the bits
    of the reduction launch it replaces, in the product and in everything the LayerNorm backward writes."""
    from protein_transformer_amd import kernels as K
    g = torch.Generator().manual_seed(T + N)
    dy = (torch.randn(T, N, generator=g) * torch.exp(torch.randn(T, 1, generator=g))).to(dev)
    w = (torch.randn(N, D, generator=g) * 0.05).to(dev)
    x, dres = (torch.randn(T, D, generator=g).to(dev) for _ in range(2))
    gam = (torch.rand(D, generator=g) + 0.5).to(dev)
    _, mean, rstd = K.layernorm_fwd(x, gam, torch.zeros(D, device=dev))
    ref = K.linear_bwd_input(dy, w, arith=K.GEMM_AUTO)
    got = K.linear_bwd_input(dy, w, arith=K.GEMM_AUTO, defer_reduce=True)
    if want == 0:                                     # enough tokens: the product is not split, nothing to defer
        assert torch.is_tensor(got) and torch.equal(got, ref)
        return
    assert isinstance(got, K.Slabs) and got.n == want and got.shape == (T, D)
    assert torch.equal(got.sum(), ref)
    outs = []
    for d in (ref, got):
        dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        rs, bs = torch.zeros(T, dtype=torch.int32, device=dev), torch.zeros(T, dtype=torch.int32, device=dev)
        mins = torch.full((2,), 0x7F000000, dtype=torch.int32, device=dev)
        planes = torch.zeros(K.lib().ptamd_hp_bytes(T, D), dtype=torch.uint8, device=dev)    # (zeros: the row padding is not written)
        dx, dr = K.layernorm_bwd_dropout(d, x, gam, mean, rstd, dg, db, dres, 0.1, 99, 5, row_scale=rs,
                                         bound_factor=torch.tensor([2.5], device=dev), bound_scale=bs, row_scale_min=mins[0:1],
                                         bound_scale_min=mins[1:2], planes=planes)
        outs.append((dx, dr, dg, db, rs, bs, mins, planes))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert torch.isfinite(outs[0][0]).all()
    with pytest.raises(RuntimeError):                 # slabs with anything of an epilogue on them are refused
        K.gemm(dy, w, None, M=T, N=D, K=N, lda=N, ldb=D, ldc=D, b_kmajor=True, split_k=want, flags=K.EPI_SLABS | K.EPI_RELU,
               ws=torch.empty(K.lib().ptamd_gemm_workspace_bytes(T, D, want), dtype=torch.uint8, device=dev))


@pytest.mark.parametrize("T,D,Kd,want,p", [(2048, 512, 2048, 4, 0.1), (4096, 512, 2048, 2, 0.1), (2048, 512, 1536, 3, 0.25),
                                           (1900, 256, 2048, 4, 0.1), (2047, 512, 2048, 4, 0.0), (16384, 512, 2048, 0, 0.1)])
def test_unreduced_k_slices_into_the_layernorm_forward(dev, T, D, Kd, want, p):
    """... and the forward twin (ptamd_layernorm_fwd_sum): FFN layer 2's split product leaves its K slices to the next
    layer's LayerNorm, which makes x = residual + dropout(sum + bias) with the epilogue's own decisions and normalises it from
    registers - the bits of reduction launch + ptamd_layernorm_fwd, in x and in everything the LayerNorm writes."""
    from protein_transformer_amd import kernels as K
    g = torch.Generator().manual_seed(T + Kd)
    a = (torch.randn(T, Kd, generator=g).clamp_min(0) * torch.exp(torch.randn(T, 1, generator=g))).to(dev)
    w = (torch.randn(D, Kd, generator=g) * 0.05).to(dev)
    bias, gam, bet = ((torch.randn(D, generator=g) * 0.3 + o).to(dev) for o in (0.0, 1.0, 0.0))
    res = torch.randn(T, D, generator=g).to(dev)
    kw = dict(residual=res, ldr=D, dropout_p=p, seed=77, stream_id=6, arith=K.GEMM_AUTO)
    x_ref = K.linear_fwd(a, w, bias, **kw)
    pend = K.linear_fwd(a, w, bias, defer_reduce=True, **kw)
    if want == 0:
        assert torch.is_tensor(pend) and torch.equal(pend, x_ref)
        return
    assert isinstance(pend, K.PendingRows) and pend.slabs.n == want and pend.value is None
    outs = []
    for x in (x_ref, pend):
        rs = torch.zeros(T, dtype=torch.int32, device=dev)
        planes = torch.zeros(K.lib().ptamd_hp_bytes(T, D), dtype=torch.uint8, device=dev)
        y, mean, rstd = K.layernorm_fwd(x, gam, bet, row_scale=rs, planes=planes)
        outs.append((x if torch.is_tensor(x) else x.value, y, mean, rstd, rs, planes))
    for u, v in zip(*outs):
        assert torch.equal(u, v)
    assert torch.isfinite(outs[0][1]).all()
    if p > 0:
        dropped = float((outs[1][0] == res).float().mean())          # (where the product was dropped the row is the residual)
        assert abs(dropped - p) < 0.01


@pytest.mark.parametrize("nprot", [4, 8])
def test_deferred_reduction_leaves_the_step_unchanged(dev, nprot, monkeypatch):
    """... and a training step with the K slices deferred == the step with the reduction launches, bit for bit (4 proteins:
    4 / 3 slices of dh2 / dh1; 8 proteins: 2 / 2)."""
    import types
    from protein_transformer_amd import kernels as K
    from protein_transformer_amd.optim import FusedSGD
    from protein_transformer_amd.train import train_step
    args = types.SimpleNamespace(loss="drmsd", combined_drmsd_weight=0.5, backbone_loss=False, clip=1.0)
    flats, seen = [], []
    real, real_fwd = K.layernorm_bwd_dropout, K.layernorm_fwd
    seen_fwd = []

    def spy(dy, *a, **kw):
        seen.append(dy.n if isinstance(dy, K.Slabs) else 1)
        return real(dy, *a, **kw)

    def spy_fwd(x, *a, **kw):
        seen_fwd.append(x.slabs.n if isinstance(x, K.PendingRows) else 1)
        return real_fwd(x, *a, **kw)
    monkeypatch.setattr(K, "layernorm_bwd_dropout", spy)
    monkeypatch.setattr(K, "layernorm_fwd", spy_fwd)
    for defer in (True, False):
        monkeypatch.setattr(K, "DEFER_REDUCE", defer)
        del seen[:], seen_fwd[:]
        m, data = _prep_model(dev, nl=3, seed=17, nprot=nprot)
        opt = FusedSGD(m, lr=1e-2, weight_decay=10e-3)
        for _ in range(2):
            train_step(m, opt, args, *data)
        flats.append(m.flat_parameters()[0].clone())
        want = ([4, 3] if nprot == 4 else [2, 2]) if defer else [1, 1]
        assert seen == [want[0], want[1], want[0], want[1], want[0]] * 2, seen   # (layer 0's dh1 goes to the unfused kernel)
        # forward: the first LayerNorm of layers 1 and 2 makes the output of the layer below (FFN layer 2: K = 2048)
        assert seen_fwd == [1, 1, want[0], 1, want[0], 1] * 2, seen_fwd
    assert torch.isfinite(flats[0]).all() and torch.equal(flats[0], flats[1])
