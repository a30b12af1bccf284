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


# ------------------------------------------------------------------------------------------------ ptamd_gemm_hp_dw
@pytest.mark.parametrize("T,M,N,split", [(64, 256, 128, 1), (512, 512, 512, None), (300, 200, 72, 3), (4096, 512, 2048, None),
                                         (2048, 1536, 512, None), (1000, 24, 512, 4), (97, 132, 48, 1), (8192, 2048, 512, 16)])
def test_gemm_hp_dw_matches_fp64(dev, T, M, N, split):
    """dW = dy^T x and dbias = column sums of dy from token-major pre-split operands (csrc/gemm_hp_dw.hip): token rows of
    very different size (per-token scales inside the contraction), a zero row, accumulation into dW / dbias."""
    from protein_transformer_amd import kernels as K_
    g = torch.Generator().manual_seed(T + M + N)
    dy = torch.randn(T, M, generator=g) * torch.exp(torch.randn(T, 1, generator=g) * 2)      # ~5 decades between tokens
    x = torch.randn(T, N, generator=g) * torch.exp(torch.randn(T, 1, generator=g) * 0.5)
    dy[T // 3] = 0
    Y, X = K_.hp_split(dy.to(dev)), K_.hp_split(x.to(dev))
    dw0, db0 = torch.randn(M, N, generator=g), torch.randn(M, generator=g)
    dw, db = dw0.to(dev), db0.to(dev)
    K_.gemm_hp_dw(Y, X, dw, db, accumulate=True, split_k=split)
    ref = dy.double().t() @ x.double()
    bound = dy.double().abs().t() @ x.double().abs()
    err = (dw.cpu().double() - dw0.double() - ref).abs()
    # norm-wise bound of a uniform-scale f16x2 product (include/ptamd.h): 2^-20 sum|x||y| + what the tokens far below the
    # largest lose; measured 2-4e-7 of the bound, plus the rounding of the accumulation into dw0
    assert float((err / (bound + dw0.double().abs())).max()) < 2e-6, float((err / (bound + dw0.double().abs())).max())
    assert float(err.norm() / ref.norm()) < 5e-7
    cref = dy.double().sum(0)
    cerr = (db.cpu().double() - db0.double() - cref).abs()
    assert float((cerr / (dy.double().abs().sum(0) + db0.double().abs())).max()) < 2e-6
    # overwrite instead of accumulate, no bias gradient
    dw2 = torch.full((M, N), float("nan"), device=dev)
    K_.gemm_hp_dw(Y, X, dw2, None, accumulate=False, split_k=split)
    assert float((dw2.cpu().double() - ref).norm() / ref.norm()) < 5e-7
    # against the uniform-scale f16x2 product of ptamd_gemm on the fp32 operands: same error class
    if M % 4 == 0 and N % 4 == 0:
        dw3 = torch.zeros(M, N, device=dev)
        K_.linear_bwd_weight(dy.to(dev), x.to(dev), dw3, None, arith=K_.GEMM_BF16X3)
        assert float((dw3.cpu().double() - ref).norm() / ref.norm()) < 5e-7


def test_gemm_hp_dw_uniform_rows_is_fp32_grade(dev):
    """Tokens of similar size (LayerNorm outputs x gradients within a few binades): element-wise fp32-fma-chain level."""
    from protein_transformer_amd import kernels as K_
    g = torch.Generator().manual_seed(5)
    T, M, N = 2048, 256, 384
    dy, x = torch.randn(T, M, generator=g), torch.randn(T, N, generator=g)
    dw = torch.zeros(M, N, device=dev)
    K_.gemm_hp_dw(K_.hp_split(dy.to(dev)), K_.hp_split(x.to(dev)), dw, None, accumulate=False)
    ref = dy.double().t() @ x.double()
    bound = dy.double().abs().t() @ x.double().abs()
    assert float(((dw.cpu().double() - ref).abs() / bound).max()) < 6e-7
