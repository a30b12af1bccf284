"""Pre-split K / V (csrc/kv_format.h): the epilogue of the QKV product (ptamd_gemm_hp, kv_planes) writes the key and value
projections as the two f16 planes + group scales the f16x2 attention kernels multiply with, the 256-query forward kernel fills
its stages from them by LDS-DMA and the one-sweep backward kernel loads its key rows from them (reference: Attention.py:49-55 -
one projection feeds Q, K and V of the scaled dot product).

  * the planes against a numpy restatement of the format (layout, swizzle, group scales, split arithmetic): byte for byte;
    the Q columns of the same call against the call without planes: bit for bit;
  * the forward kernel on planes == the forward kernel on the fp32 K / V they were made from, bit for bit (output, log-sum-exp
    and the exported dropout decisions): same scaling groups, same split;
  * the backward kernel on planes against the fp32 one (a key row is scaled with its group of four there: equal to rounding)
    and both against dense fp64 attention;
  * a model step with and without the planes.
"""
import os

import numpy as np
import pytest
import torch

from test_gpu_kernels import assert_close, ref_attention

pytestmark = pytest.mark.gpu

B, L, H, DK = 17, 256, 8, 64          # 17 x 8 (protein, head) pairs: the 256-query forward kernel and the one-sweep backward kernel
D = H * DK
T = B * L


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rev3(x):
    return ((x & 1) << 2) | (x & 2) | ((x >> 2) & 1)


def decode_planes(planes, inv, T, H):
    """kv_format.h in numpy: -> (hi, lo) float16 [2 (K, V), H, T, 64] in natural order and inverse scales [2, H, T]."""
    nt = T // 32
    raw = planes.cpu().numpy().view(np.float16).reshape(2, H, nt, 2, 32, 8, 8)      # [which, h, tile, plane, row, chunk position, 8]
    out = np.empty((2, H, nt, 2, 32, 8, 8), np.float16)
    for r in range(32):
        for c in range(8):
            out[:, :, :, :, r, c] = raw[:, :, :, :, r, c ^ rev3((r >> 1) & 7)]
    out = out.reshape(2, H, nt, 2, 32, 64)
    hi = out[:, :, :, 0].reshape(2, H, T, 64)
    lo = out[:, :, :, 1].reshape(2, H, T, 64)
    iv = inv.cpu().numpy().reshape(2, H, nt, 8)
    slot = np.array([(g & 1) * 4 + (g >> 1) for g in range(8)])
    inv_tok = np.repeat(iv[:, :, :, slot], 4, axis=3).reshape(2, H, T)              # group g = row >> 2 of the tile
    return hi, lo, inv_tok


def scale_of(amax):
    amax = np.asarray(amax, np.float32)
    e = (amax.view(np.uint32) >> 23).astype(np.int64)
    return np.ldexp(1.0, np.minimum(268 - e, 254) - 127).astype(np.float32)


def make_qkv(dev, seed=0, with_kv=True):
    """x W^T + b through ptamd_gemm_hp: -> (fp32 qkv of the call without planes, fp32 qkv of the call with planes, kv buffers)."""
    from protein_transformer_amd import kernels as K
    g = torch.Generator().manual_seed(seed)
    # (projections of the size the dense-fp64 tests of tests/test_gpu_kernels.py use: |q|, |k|, |v| ~ 1.2, a few tokens 2-3 x larger)
    x = (torch.randn(T, D, generator=g) * torch.exp(0.25 * torch.randn(T, 1, generator=g))).to(dev)
    w = (torch.randn(3 * D, D, generator=g) / np.sqrt(D) * 1.1).to(dev)
    bias = (torch.randn(3 * D, generator=g) * 0.3).to(dev)
    a, bop = K.hp_split(x), K.hp_split(w)
    ref = K.gemm_hp(a, bop, torch.empty(T, 3 * D, device=dev), bias=bias)
    if not with_kv:
        return ref, None, None
    kv = K.attention_kv_buffers(T, H, dev)
    kv[0].fill_(0x5A)
    got = K.gemm_hp(a, bop, torch.full((T, 3 * D), float("nan"), device=dev), bias=bias, kv=kv, kv_col0=D, kv_heads=H)
    torch.cuda.synchronize()
    return ref, got, kv


def test_qkv_epilogue_writes_the_planes(dev):
    ref, got, kv = make_qkv(dev, seed=1)
    assert torch.equal(got[:, :D], ref[:, :D])                       # Q: the usual fp32 store
    assert torch.isnan(got[:, D:]).all()                             # K | V: not written as fp32 at all
    hi, lo, inv = decode_planes(kv[0], kv[1], T, H)
    x = ref[:, D:].cpu().numpy().reshape(T, 2, H, 64).transpose(1, 2, 0, 3)           # [which, h, T, 64]
    amax = np.abs(x).reshape(2, H, T // 4, 4 * 64).max(-1)                            # groups of four tokens, all 64 d of a head
    s = np.repeat(scale_of(amax), 4, axis=2)                                          # [2, H, T]
    assert np.array_equal(inv, (1.0 / s.astype(np.float64)).astype(np.float32))
    xs = x * s[..., None]                                                             # exact: powers of two
    want_hi = xs.astype(np.float16)
    want_lo = (xs - want_hi.astype(np.float32)).astype(np.float16)
    assert np.array_equal(hi.view(np.uint16), want_hi.view(np.uint16))
    assert np.array_equal(lo.view(np.uint16), want_lo.view(np.uint16))
    top = np.abs(xs).reshape(2, H, T // 4, 256).max(-1)
    assert np.all((top >= 2.0 ** 14) & (top < 2.0 ** 15))


def _seq(lens, seed=0):
    seq = torch.full((B, L), 20, dtype=torch.int64)
    for b, n in enumerate(lens):
        seq[b, :n] = torch.randint(0, 20, (n,), generator=torch.Generator().manual_seed(seed + b))
    return seq


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_attention_on_planes(dev, p):
    from protein_transformer_amd import kernels as K
    for ar in (K.GEMM_AUTO, K.GEMM_F16X2):
        assert K.attention_reads_kv_planes(B, L, H, DK, ar)
    assert not K.attention_reads_kv_planes(B, L, H, DK, K.GEMM_BF16X3) and not K.attention_reads_kv_planes(B, L + 1, H, DK, K.GEMM_AUTO)
    assert not K.attention_reads_kv_planes(2, L, H, DK, K.GEMM_AUTO) and not K.attention_reads_kv_planes(B, L, H, 32, K.GEMM_AUTO)
    ref, got, kv = make_qkv(dev, seed=2)
    lens = [L] * 11 + [200, 97, 1, 256, 33, 160]
    seq = _seq(lens).to(dev)
    seed, sid = 777, 3
    bits_a = K.attention_keep_bits(B, L, H, dev) if p > 0 else None
    bits_b = K.attention_keep_bits(B, L, H, dev) if p > 0 else None
    o_ref, lse_ref = K.attention_fwd(ref, seq, H, p, seed, sid, arith=K.GEMM_AUTO, keep_bits=bits_a)
    o_kv, lse_kv = K.attention_fwd(got, seq, H, p, seed, sid, arith=K.GEMM_AUTO, keep_bits=bits_b, kv=kv)
    torch.cuda.synchronize()
    assert torch.isfinite(o_kv).all()
    assert torch.equal(o_kv, o_ref) and torch.equal(lse_kv, lse_ref)                   # same groups, same split: the same bits
    if p > 0:
        assert torch.equal(bits_a, bits_b)
    g = torch.Generator().manual_seed(5)
    dout = torch.randn(T, D, generator=g).to(dev)
    d_ref = K.attention_bwd(ref, seq, o_ref, dout, lse_ref, H, p, seed, sid, arith=K.GEMM_AUTO, keep_bits=bits_a)
    d_kv = K.attention_bwd(got, seq, o_kv, dout, lse_kv, H, p, seed, sid, arith=K.GEMM_AUTO, keep_bits=bits_b, kv=kv)
    again = K.attention_bwd(got, seq, o_kv, dout, lse_kv, H, p, seed, sid, arith=K.GEMM_AUTO, keep_bits=bits_b, kv=kv)
    torch.cuda.synchronize()
    assert torch.equal(d_kv, again)
    assert torch.isfinite(d_kv).all()
    assert ((d_kv - d_ref).norm() / d_ref.norm()).item() < 1e-6
    if p == 0.0:      # dense fp64 attention on the same fp32 projections
        q64 = ref.double().cpu().view(B, L, 3 * D).requires_grad_()
        out64, _ = ref_attention(q64, seq.cpu() != 20, H)
        out64.backward(dout.double().cpu().view(B, L, D))
        valid = (seq.cpu() != 20).view(T)
        # (the bar of tests/test_gpu_kernels.py on its uniform +-1.5 inputs is 1e-5 / 2e-6; these projections have a few tokens 2-3 x
        # larger - the fp32-input kernel, bit-identical above, sits at the same 5e-6)
        assert_close(o_kv[valid.to(dev)], out64.detach().view(T, D)[valid], 2e-5, 4e-6, "attention fwd on planes")
        want = q64.grad.view(T, 3 * D)
        assert_close(d_kv, want, 1e-4, 2e-6 * max(1.0, want.abs().max().item()), "attention bwd on planes")
    # refused where the kernels that read planes would not run
    small = K.attention_kv_buffers(2 * L, H, dev)
    with pytest.raises(RuntimeError):
        K.attention_fwd(ref[:2 * L], seq[:2], H, 0.0, 0, 0, arith=K.GEMM_AUTO, kv=small)
    with pytest.raises(RuntimeError):
        K.attention_fwd(ref, seq, H, 0.0, 0, 0, arith=K.GEMM_BF16X3, kv=kv)


def test_model_step_with_and_without_planes(dev):
    """d512 / 2 layers / 17 x 256: gradients of a guard-trusted training pass with K / V as planes against the same pass with fp32
    K / V - forward identical (same bits), backward equal to rounding - and the planes really in use."""
    import types
    from protein_transformer_amd import synthetic
    from protein_transformer_amd.models.encoder_only import EncoderOnlyTransformer
    from protein_transformer_amd.protein.Sequence import VOCAB
    from protein_transformer_amd.protein.Structure import nerf_forward
    from protein_transformer_amd.train import get_losses
    build = lambda ang, seq: nerf_forward(ang.to(dev), seq.to(dev))[0]  # noqa: E731
    batch = synthetic.make_batch([L] * 13 + [200, 97, 31, 160], L_pad=L, seed=9, build_coords=build)
    seq, ang, crd = (batch[k].to(dev) for k in ("seq", "true_ang", "true_crd"))
    args = types.SimpleNamespace(loss="drmsd", combined_drmsd_weight=0.5, backbone_loss=False, clip=None)
    res = {}
    for planes in (True, False):
        torch.manual_seed(3)
        m = EncoderOnlyTransformer(2, H, D, 2048, L, VOCAB, synthetic.angle_means(batch["true_ang"]), True, dropout=0.1).to(dev).train()
        with torch.no_grad():
            dict(m.named_parameters())["output_projection.weight"].normal_(0, 0.02)
        m.kv_planes = planes
        for it in range(2):
            m.zero_grad()
            pred = m(seq, ang)
            losses = get_losses(args, pred, ang, crd, seq)
            m.auto_guard.settle()
        res[planes] = (pred.detach().clone(), m.flat_parameters()[1].clone(), float(losses["drmsd-full"]),
                       m.__dict__.get("_kv_plane_passes", 0))
    assert res[True][3] == 2 and res[False][3] == 0          # the second (guard-trusted) pass of both layers; never without the flag
    assert torch.equal(res[True][0], res[False][0]) and res[True][2] == res[False][2]
    ga, gb = res[True][1], res[False][1]
    assert ((ga - gb).norm() / gb.norm()).item() < 1e-5


def test_planes_under_the_key_block_split_sweep(dev):
    """Round 6: 16 proteins x 8 heads x 512 - 128 (protein, head) pairs, the per-GPU share at two GPUs - run the 256-query forward
    kernel and the one-sweep backward kernel split per 256-key block: both read pre-split K / V there too.  Forward the same bits
    as on fp32 K / V, backward equal to rounding and reproducible; 8 proteins (query ranges as well) keep fp32 K / V."""
    from protein_transformer_amd import kernels as K
    B2, L2 = 16, 512
    T2 = B2 * L2
    assert K.attention_reads_kv_planes(B2, L2, H, DK, K.GEMM_AUTO)
    assert not K.attention_reads_kv_planes(8, L2, H, DK, K.GEMM_AUTO) and not K.attention_reads_kv_planes(4, L2, H, DK, K.GEMM_AUTO)
    g = torch.Generator().manual_seed(12)
    x = (torch.randn(T2, D, generator=g) * torch.exp(0.25 * torch.randn(T2, 1, generator=g))).to(dev)
    w = (torch.randn(3 * D, D, generator=g) / np.sqrt(D) * 1.1).to(dev)
    bias = (torch.randn(3 * D, generator=g) * 0.3).to(dev)
    a, bop = K.hp_split(x), K.hp_split(w)
    ref = K.gemm_hp(a, bop, torch.empty(T2, 3 * D, device=dev), bias=bias)
    kv = K.attention_kv_buffers(T2, H, dev)
    got = K.gemm_hp(a, bop, torch.full((T2, 3 * D), float("nan"), device=dev), bias=bias, kv=kv, kv_col0=D, kv_heads=H)
    seq = torch.full((B2, L2), 20, dtype=torch.int64)
    for b, n in enumerate([L2] * 12 + [300, 97, 512, 33]):
        seq[b, :n] = torch.randint(0, 20, (n,), generator=torch.Generator().manual_seed(b))
    seq = seq.to(dev)
    p, seed, sid = 0.1, 31, 2
    bits_a, bits_b = K.attention_keep_bits(B2, L2, H, dev), K.attention_keep_bits(B2, L2, H, dev)
    o_ref, lse_ref = K.attention_fwd(ref, seq, H, p, seed, sid, arith=K.GEMM_AUTO, keep_bits=bits_a)
    o_kv, lse_kv = K.attention_fwd(got, seq, H, p, seed, sid, arith=K.GEMM_AUTO, keep_bits=bits_b, kv=kv)
    assert torch.equal(o_kv, o_ref) and torch.equal(lse_kv, lse_ref) and torch.equal(bits_a, bits_b)
    dout = torch.randn(T2, D, generator=g).to(dev)
    d_ref = K.attention_bwd(ref, seq, o_ref, dout, lse_ref, H, p, seed, sid, arith=K.GEMM_AUTO, keep_bits=bits_a)
    d_kv = K.attention_bwd(got, seq, o_kv, dout, lse_kv, H, p, seed, sid, arith=K.GEMM_AUTO, keep_bits=bits_b, kv=kv)
    again = K.attention_bwd(got, seq, o_kv, dout, lse_kv, H, p, seed, sid, arith=K.GEMM_AUTO, keep_bits=bits_b, kv=kv)
    torch.cuda.synchronize()
    assert torch.isfinite(d_kv).all() and torch.equal(d_kv, again)
    assert ((d_kv - d_ref).norm() / d_ref.norm()).item() < 1e-6
