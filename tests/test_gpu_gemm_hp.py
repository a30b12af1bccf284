"""MI355X tests of the pre-split ("half-pair") operand format and ptamd_gemm_hp (csrc/hp_format.h, csrc/gemm_hp.hip).

The arithmetic is PTAMD_GEMM_F16X2 of include/ptamd.h (two row-scaled f16 terms, three MFMA products, f32 accumulate):
fp32-grade against fp64 on operands of moderate dynamic range, norm-wise on wide ones - the same assertions as
tests/test_gpu_kernels.py::test_gemm_f16x2_error_model makes for the kernel that splits while it stages.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def unpack(op, dev):
    """HpOperand -> (hi + lo) / scale as fp64 [rows, K] on the host, undoing the block / chunk layout with numpy."""
    rows_p, Kp = (op.rows + 31) // 32 * 32, (op.K + 31) // 32 * 32
    raw = op.planes.cpu().numpy().view(np.float16).reshape(rows_p // 32, Kp // 16, 2, 64, 8)   # [rb][kb][plane][chunk][8]
    out = np.zeros((2, rows_p, Kp))
    r = np.arange(32)
    for h in (0, 1):
        c = 2 * r + (h ^ ((r >> 3) & 1))
        blk = raw[:, :, :, c, :]                                                                # [rb][kb][plane][r][8]
        for p in (0, 1):
            v = blk[:, :, p].astype(np.float64)                                                 # [rb][kb][r][8]
            v = v.transpose(0, 2, 1, 3)                                                         # [rb][r][kb][8]
            tmp = np.zeros((rows_p // 32, 32, Kp // 16, 16))
            tmp[..., 8 * h:8 * h + 8] = v
            out[p] += tmp.reshape(rows_p, Kp) * 1.0
    scale = op.scale.cpu().numpy().astype(np.float64)
    return (out[0] + out[1]) / scale[:, None], out, scale


@pytest.mark.parametrize("rows,K,transposed", [(64, 32, False), (100, 48, False), (33, 16, False), (256, 512, False),
                                               (48, 64, True), (512, 2048, True), (130, 24, False)])
def test_hp_split_roundtrip(dev, rows, K, transposed):
    from protein_transformer_amd import kernels as K_
    g = torch.Generator().manual_seed(rows * 7 + K)
    x = torch.randn(rows, K, generator=g) * torch.exp(torch.randn(rows, 1, generator=g) * 3)   # rows of very different size
    x[rows // 2] = 0                                                                            # a row of zeros
    src = x.t().contiguous() if transposed else x
    op = K_.hp_split(src.to(dev), transposed=transposed)
    back, planes, scale = unpack(op, dev)
    assert np.all(planes[:, rows:, :] == 0) and np.all(planes[:, :, K:] == 0)                  # zero padding
    assert np.all(scale[rows:] == 1.0)
    amax = x.abs().max(1).values.double().numpy()
    nz = amax > 0
    assert np.all(np.log2(scale[:rows][nz]) == np.round(np.log2(scale[:rows][nz])))           # powers of two
    top = amax[nz] * scale[:rows][nz]
    assert np.all((top >= 2.0 ** 14) & (top < 2.0 ** 15))
    err = np.abs(back[:rows, :K] - x.double().numpy())
    assert np.all(err <= 2.0 ** -22 * np.abs(x.double().numpy()) + 2.0 ** -39 * amax[:, None] + 1e-300)
    assert np.all(back[rows // 2] == 0)


@pytest.mark.parametrize("M,N,K", [(256, 128, 32), (512, 512, 512), (300, 200, 64), (1000, 1536, 512), (640, 24, 512),
                                   (2048, 512, 2048), (77, 130, 48)])
def test_gemm_hp_matches_fp64(dev, M, N, K):
    from protein_transformer_amd import kernels as K_
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))
    b = torch.randn(N, K, generator=g) * 0.05
    C = torch.full((M, N), float("nan"), device=dev)
    K_.gemm_hp(K_.hp_split(a.to(dev)), K_.hp_split(b.to(dev)), C)
    ref = a.double() @ b.double().t()
    bound = (a.double().abs() @ b.double().abs().t())
    err = (C.cpu().double() - ref).abs()
    assert float((err / bound).max()) < 6e-7                       # fp32-fma-chain level (measured ~3e-7)
    # identical to the kernel that splits while it stages, up to the order of the f32 accumulation
    C2 = torch.empty(M, N, device=dev)
    K_.gemm(a.to(dev), b.to(dev), C2, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, arith=K_.GEMM_F16X2) if K % 4 == 0 and K >= 16 else None
    if K % 4 == 0 and K >= 16:
        assert float(((C2 - C).abs().cpu().double() / bound).max()) < 3e-7


def test_gemm_hp_epilogues_split_and_masks(dev):
    """bias / ReLU / dropout / residual / gate / tanh / accumulate and split-K give what ptamd_gemm gives (same dropout
    masks: the generator is indexed by (row, column), not by the kernel's tile shape)."""
    from protein_transformer_amd import kernels as K_
    g = torch.Generator().manual_seed(3)
    M, N, K = 520, 384, 256
    a, b = torch.randn(M, K, generator=g).to(dev), (torch.randn(N, K, generator=g) * 0.1).to(dev)
    bias, res = torch.randn(N, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev)
    A, B = K_.hp_split(a), K_.hp_split(b)

    def both(**epi):
        c1 = torch.randn(M, N, generator=torch.Generator().manual_seed(9)).to(dev)
        c2 = c1.clone()
        K_.gemm_hp(A, B, c1, **epi)
        K_.gemm(a, b, c2, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, arith=K_.GEMM_F16X2, **epi)
        return c1, c2

    for epi in (dict(bias=bias), dict(bias=bias, flags=K_.EPI_RELU), dict(bias=bias, residual=res, ldr=N),
                dict(bias=bias, flags=K_.EPI_TANH), dict(flags=K_.EPI_ACCUM),
                dict(residual=res, ldr=N, flags=K_.EPI_GATE, gate_scale=1.25),
                dict(bias=bias, flags=K_.EPI_RELU, dropout_p=0.3, seed=77, stream_id=5),
                dict(bias=bias, residual=res, ldr=N, dropout_p=0.1, seed=78, stream_id=6)):
        c1, c2 = both(**epi)
        assert torch.equal(c1 == 0, c2 == 0), epi                     # same ReLU / dropout / gate pattern
        assert float((c1 - c2).abs().max()) < 2e-5 * float(c2.abs().max()), epi
    c1 = torch.zeros(M, N, device=dev)
    K_.gemm_hp(A, B, c1, split_k=4, bias=bias)
    c2 = torch.zeros(M, N, device=dev)
    K_.gemm_hp(A, B, c2, bias=bias)
    assert float((c1 - c2).abs().max()) < 2e-5 * float(c2.abs().max())


def test_layernorm_writes_planes_and_weights_split_in_one_launch(dev):
    """The two writers behind the hp forward products of the model: ptamd_layernorm_fwd(planes=...) and ptamd_hp_split_rows
    produce exactly what ptamd_hp_split makes of the same fp32 matrix, and the product from them equals the fp64 product."""
    from protein_transformer_amd import kernels as K_
    g = torch.Generator().manual_seed(11)
    T, D, N = 200, 96, 160                                   # T not a multiple of 32: the last block row is partly unused
    x = (torch.randn(T, D, generator=g) * torch.exp(torch.randn(T, 1, generator=g))).to(dev)
    gamma, beta = (1 + 0.2 * torch.randn(D, generator=g)).to(dev), (0.1 * torch.randn(D, generator=g)).to(dev)
    scale = torch.empty(T, dtype=torch.int32, device=dev)
    planes = torch.zeros(K_.lib().ptamd_hp_bytes(T, D), dtype=torch.uint8, device=dev)
    y, _, _ = K_.layernorm_fwd(x, gamma, beta, row_scale=scale, planes=planes)
    y2, _, _ = K_.layernorm_fwd(x, gamma, beta)
    assert torch.equal(y, y2)
    ref_op = K_.hp_split(y)
    got = K_.hp_view(planes, scale, T, D)
    assert torch.equal(got.scale[:T], ref_op.scale[:T])
    back_ref, _, _ = unpack(ref_op, dev)
    back_got = unpack(K_.hp_view(planes, torch.cat([got.scale, torch.ones(24, device=dev)]), T, D), dev)[0]
    assert np.array_equal(back_got[:T], back_ref[:T])
    # weights: several matrices, one launch
    ws = [torch.randn(N, D, generator=g).to(dev) * 0.05, torch.randn(64, D, generator=g).to(dev), torch.randn(40, 32, generator=g).to(dev)]
    outs = K_.hp_split_rows(ws, [K_.HpOperand(w.shape[0], w.shape[1], dev) for w in ws])
    for w, o in zip(ws, outs):
        r = K_.hp_split(w)
        assert torch.equal(o.planes, r.planes) and torch.equal(o.scale, r.scale)
    C = torch.empty(T, N, device=dev)
    K_.gemm_hp(got, outs[0], C)
    ref = y.cpu().double() @ ws[0].cpu().double().t()
    bound = y.cpu().double().abs() @ ws[0].cpu().double().abs().t()
    assert float(((C.cpu().double() - ref).abs() / bound).max()) < 6e-7


def test_backward_writers_of_the_format(dev):
    """Round 4: the two writers behind the hp dX product of FFN layer 2 - ptamd_layernorm_bwd_dropout(planes=...) (the
    dropped gradient rows, with the row scales it computes anyway) and ptamd_hp_split_cols (the transposed weights, with the
    column scales of ptamd_weight_scales) - produce exactly what ptamd_hp_split makes of the same fp32 matrices, and the gated
    product from them equals the staging kernel's."""
    from protein_transformer_amd import kernels as K_
    g = torch.Generator().manual_seed(12)
    T, D, F, p, seed, sid = 200, 96, 160, 0.1, 77, 3
    x = torch.randn(T, D, generator=g).to(dev)
    dy = (torch.randn(T, D, generator=g) * torch.exp(torch.randn(T, 1, generator=g))).to(dev)
    dres = torch.randn(T, D, generator=g).to(dev)
    gamma = (1 + 0.2 * torch.randn(D, generator=g)).to(dev)
    _, mean, rstd = K_.layernorm_fwd(x, gamma, torch.zeros(D, device=dev))
    dgam, dbet = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    scale = torch.empty(T, dtype=torch.int32, device=dev)
    planes = torch.zeros(K_.lib().ptamd_hp_bytes(T, D), dtype=torch.uint8, device=dev)
    dx, dropped = K_.layernorm_bwd_dropout(dy, x, gamma, mean, rstd, dgam, dbet, dres, p, seed, sid, row_scale=scale, planes=planes)
    dx2, dropped2 = K_.layernorm_bwd_dropout(dy, x, gamma, mean, rstd, torch.zeros(D, device=dev), torch.zeros(D, device=dev), dres,
                                             p, seed, sid)
    assert torch.equal(dx, dx2) and torch.equal(dropped, dropped2)                 # the by-product does not touch the results
    ref_op = K_.hp_split(dropped)
    got = K_.hp_view(planes, scale, T, D)
    assert torch.equal(got.scale[:T], ref_op.scale[:T])
    back_ref = unpack(ref_op, dev)[0]
    back_got = unpack(K_.hp_view(planes, torch.cat([got.scale, torch.ones(24, device=dev)]), T, D), dev)[0]
    assert np.array_equal(back_got[:T], back_ref[:T])
    # W [D, F] -> planes of W^T [F, D] with the column scales of ptamd_weight_scales
    w = (torch.randn(D, F, generator=g) * 0.05 * torch.exp(torch.randn(1, F, generator=g))).to(dev)
    cs = torch.empty(F, dtype=torch.int32, device=dev)
    K_.weight_scales([dict(w=w, col_scale=cs)])
    out = K_.hp_view(torch.zeros(K_.lib().ptamd_hp_bytes(F, D), dtype=torch.uint8, device=dev), cs, F, D)
    K_.hp_split_cols([w], [cs], [out])
    r = K_.hp_split(w, transposed=True)
    assert torch.equal(out.scale[:F], r.scale[:F]) and torch.equal(out.planes, r.planes)
    # dz = gate(dropped W): LDS-DMA kernel from the two writers against the staging kernel on the fp32 operands
    f1 = torch.relu(torch.randn(T, F, generator=g)).to(dev)
    a = K_.gemm_hp(got, out, torch.empty(T, F, device=dev), residual=f1, ldr=F, flags=K_.EPI_GATE, gate_scale=1.0 / (1.0 - p))
    b = K_.linear_bwd_input(dropped, w, gate=f1, gate_dropout_p=p, arith=K_.GEMM_F16X2)
    assert torch.equal(a == 0, b == 0)
    assert float((a - b).abs().max()) <= 3e-6 * float(b.abs().max())


def _gate_mask_restated(y):
    """(y > 0) of an [M, N] matrix in the layout of ptamd_gate_mask_bytes: uint64 entry ((cb * ceil(M / 32) + rb) * 16 + r),
    bit l = element (row 32 rb + (r & 3) + 8 (r >> 2) + 4 (l >> 5), column 32 cb + (l & 31))."""
    M, N = y.shape
    nrb, ncb = (M + 31) // 32, (N + 31) // 32
    pad = np.zeros((nrb * 32, ncb * 32), dtype=bool)
    pad[:M, :N] = y > 0
    blk = pad.reshape(nrb, 32, ncb, 32).transpose(2, 0, 1, 3)                       # [cb][rb][row][col]
    out = np.zeros((ncb, nrb, 16), dtype=np.uint64)
    for r in range(16):
        for half in (0, 1):
            row = (r & 3) + 8 * (r >> 2) + 4 * half
            bits = (blk[:, :, row, :].astype(np.uint64) << (np.arange(32, dtype=np.uint64) + np.uint64(32 * half))).sum(-1)
            out[:, :, r] |= bits.astype(np.uint64)
    return out.reshape(-1)


@pytest.mark.parametrize("M,N,K", [(512, 256, 64), (1000, 520, 96), (256, 128, 32), (16384, 2048, 512)])
def test_one_bit_gate_written_by_the_product_and_read_by_the_gated_one(dev, M, N, K):
    """ptamd_gemm_hp leaves `result > 0` of its ReLU + dropout output as one bit per element (gate_mask_out); the gated dX
    products (ptamd_gemm in f16x2 and bf16x3 arithmetic, ptamd_gemm_hp) reading that mask give the SAME BITS as the ones
    reading the fp32 activation; the mask is the restated layout of include/ptamd.h; padding blocks are in range."""
    from protein_transformer_amd import kernels as K_
    g = torch.Generator().manual_seed(M + N + K)
    x, w, bias = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1, torch.randn(N, generator=g) * 0.1
    A, B = K_.hp_split(x.to(dev)), K_.hp_split(w.to(dev))
    f1 = torch.empty(M, N, device=dev)
    f1b = torch.empty(M, N, device=dev)
    n64 = K_.lib().ptamd_gate_mask_bytes(M, N) // 8
    buf = torch.full((n64 + 512,), 0x1357246813572468, dtype=torch.int64, device=dev)
    kw = dict(bias=bias.to(dev), flags=K_.EPI_RELU, dropout_p=0.1, seed=11, stream_id=3)
    K_.gemm_hp(A, B, f1, gate_mask_out=buf[:n64], **kw)
    K_.gemm_hp(A, B, f1b, **kw)
    assert torch.equal(f1, f1b)                                             # the export does not touch the product
    assert (buf[n64:] == 0x1357246813572468).all()                           # ... nor anything behind the buffer
    want = _gate_mask_restated(f1.cpu().numpy())
    inside = _gate_mask_restated(np.ones((M, N), dtype=np.float32))      # (bits of rows / columns past the matrix: unspecified)
    got = buf[:n64].cpu().numpy().view(np.uint64) & inside
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {n64} mask entries differ"
    # the gated products: dz[M, N] = (dy[M, D] W2[D, N]) * (f1 > 0) / (1 - p)
    D = 64
    dy, w2 = torch.randn(M, D, generator=g).to(dev), (torch.randn(D, N, generator=g) * 0.1).to(dev)
    mask = buf[:n64]
    for ar in (K_.GEMM_F16X2, K_.GEMM_AUTO, K_.GEMM_BF16X3):
        a = K_.linear_bwd_input(dy, w2, gate=f1, gate_dropout_p=0.1, arith=ar)
        if K_.pick_split_k_rows(M, N, D) == 1 and ar != K_.GEMM_BF16X3:     # (bf16x3: the wrapper gates by f1)
            b = K_.gemm(dy, w2, torch.empty(M, N, device=dev), M=M, N=N, K=D, lda=D, ldb=N, ldc=N, b_kmajor=True,
                        flags=K_.EPI_GATE, gate_mask=mask, gate_scale=1.0 / 0.9, arith=ar)
            assert torch.equal(a, b), ar
        c = K_.linear_bwd_input(dy, w2, gate=f1, gate_dropout_p=0.1, arith=ar, gate_mask=mask)   # (falls back where it must)
        assert torch.equal(a, c), ar
    ref = (dy.double() @ w2.double()) * (f1 > 0) / 0.9
    assert ((a.double() - ref).abs().max() / ref.abs().max()).item() < 1e-5
    Ady, Bw2 = K_.hp_split(dy), K_.hp_split(w2, transposed=True)
    h0 = K_.gemm_hp(Ady, Bw2, torch.empty(M, N, device=dev), residual=f1, ldr=N, flags=K_.EPI_GATE, gate_scale=1.0 / 0.9)
    h1 = K_.gemm_hp(Ady, Bw2, torch.empty(M, N, device=dev), gate_mask=mask, flags=K_.EPI_GATE, gate_scale=1.0 / 0.9)
    assert torch.equal(h0, h1)


def test_one_bit_gate_argument_checks(dev):
    from protein_transformer_amd import kernels as K_
    M, N, D = 64, 64, 32
    dy, w2, f1 = torch.randn(M, D).to(dev), torch.randn(D, N).to(dev), torch.randn(M, N).to(dev)
    mask = K_.gate_mask_buffer(M, N, dev)
    mask.zero_()
    out = torch.empty(M, N, device=dev)
    base = dict(M=M, N=N, K=D, lda=D, ldb=N, ldc=N, b_kmajor=True, gate_scale=1.0)
    with pytest.raises(RuntimeError):                                        # both gates
        K_.gemm(dy, w2, out, flags=K_.EPI_GATE, gate_mask=mask, residual=f1, ldr=N, arith=K_.GEMM_F16X2, **base)
    with pytest.raises(RuntimeError):                                        # a mask without the flag
        K_.gemm(dy, w2, out, flags=0, gate_mask=mask, arith=K_.GEMM_F16X2, **base)
    for ar in (K_.GEMM_F32, K_.GEMM_BF16X3, K_.GEMM_BF16X3_FULL):
        with pytest.raises(RuntimeError):                                    # the f16x2 kernels only
            K_.gemm(dy, w2, out, flags=K_.EPI_GATE, gate_mask=mask, arith=ar, **base)
    with pytest.raises(RuntimeError):                                        # with a dropout of its own
        K_.gemm(dy, w2, out, flags=K_.EPI_GATE, gate_mask=mask, dropout_p=0.1, arith=K_.GEMM_F16X2, **base)
    z = K_.gemm(dy, w2, out, flags=K_.EPI_GATE, gate_mask=mask, arith=K_.GEMM_F16X2, **base)
    assert (z == 0).all()                                                    # an all-closed gate
