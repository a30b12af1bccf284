"""The fused attention backward kernel (csrc/attention_f16x2.hip: attn_bwd_fused_f16x2_kernel - dQ, dK and dV in one sweep
over the score matrix, reference Attention.py:14-22) against dense fp64 attention and against the two-kernel path.

The library picks the fused kernel when (proteins x heads) workgroups fill the chip; `PTAMD_ATTN_FUSED=1 / 0` in the
environment (read at every call) forces / forbids it for head size 64 - the tests run it on small batches that way, on
one, two and three 256-key blocks (the outer loop: dQ is stored by the first block and read - add - stored by the others),
ragged lengths, padding, dropout, wide dynamic ranges, degenerate rows, and check the row scales it leaves behind.
"""
import os

import numpy as np
import pytest
import torch

from test_gpu_kernels import assert_close, ref_attention, rnd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


class fused:
    """PTAMD_ATTN_FUSED for the calls inside: True / 1 = the unsplit one-sweep kernel, False / 0 = the two-kernel path, 2 = the
    split sweep (round 6: a workgroup per (pair, key block[, query range]) + the slab reduction)."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.old = os.environ.get("PTAMD_ATTN_FUSED")
        os.environ["PTAMD_ATTN_FUSED"] = "2" if self.on == 2 else ("1" if self.on else "0")

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("PTAMD_ATTN_FUSED", None)
        else:
            os.environ["PTAMD_ATTN_FUSED"] = self.old


def _seq(B, L, lens, seed=0):
    seq = torch.full((B, L), 20, dtype=torch.int64)
    for b, n in enumerate(lens):
        seq[b, :n] = torch.randint(0, 20, (n,), generator=torch.Generator().manual_seed(seed + b))
    return seq


@pytest.mark.parametrize("B,L,H,lens", [(2, 512, 8, [512, 300]), (1, 130, 2, [130]), (3, 700, 2, [700, 513, 31]),
                                        (2, 256, 4, [256, 1]), (1, 257, 1, [257]), (2, 33, 2, [33, 7])])
@pytest.mark.parametrize("arith", ["f16x2", "auto"])
def test_fused_backward_vs_fp64(dev, B, L, H, lens, arith):
    from protein_transformer_amd import kernels as K_
    ar = K_.GEMM_F16X2 if arith == "f16x2" else K_.GEMM_AUTO
    dk, D = 64, 64 * H
    seq = _seq(B, L, lens)
    qkv = rnd((B, L, 3 * D), 20, 1.5).double().requires_grad_()
    out, _ = ref_attention(qkv, seq != 20, H)
    dout = rnd((B, L, D), 21).double()
    out.backward(dout)
    qd = qkv.detach().float().view(B * L, 3 * D).to(dev)
    o, lse = K_.attention_fwd(qd, seq.to(dev), H, 0.0, 0, 0, arith=ar)
    with fused(True):
        dqkv = K_.attention_bwd(qd, seq.to(dev), o, dout.float().view(B * L, D).to(dev), lse, H, 0.0, 0, 0, arith=ar)
        again = K_.attention_bwd(qd, seq.to(dev), o, dout.float().view(B * L, D).to(dev), lse, H, 0.0, 0, 0, arith=ar)
    with fused(False):
        two = K_.attention_bwd(qd, seq.to(dev), o, dout.float().view(B * L, D).to(dev), lse, H, 0.0, 0, 0, arith=ar)
    ref = qkv.grad.view(B * L, 3 * D)
    assert_close(dqkv, ref, 1e-4, 2e-6 * max(1.0, ref.abs().max().item()), "fused attention bwd")
    assert torch.equal(dqkv, again)                                  # fixed summation order: bit-reproducible
    # the same dK / dV arithmetic as the two-kernel path, dQ by another route: equal to rounding
    assert ((dqkv - two).norm() / two.norm()).item() < 1e-6
    for i, name in enumerate(("dQ", "dK", "dV")):
        a, r = dqkv.view(B * L, 3, D)[:, i].double().cpu(), ref.view(B * L, 3, D)[:, i]
        assert ((a - r).norm() / r.norm()).item() < 2e-6, name


@pytest.mark.parametrize("B,L,H,lens", [(2, 512, 8, [512, 300]), (4, 512, 8, [512, 411, 77, 512]), (8, 512, 8, [512] * 8),
                                        (16, 512, 8, [512] * 15 + [129]), (1, 130, 2, [130]), (3, 700, 2, [700, 513, 31]),
                                        (2, 256, 4, [256, 1]), (1, 257, 1, [257]), (2, 33, 2, [33, 7]), (2, 1500, 8, [1500, 611])])
@pytest.mark.parametrize("p", [0.0, 0.25])
def test_split_sweep(dev, B, L, H, lens, p):
    """Round 6: the one-sweep kernel cut into a workgroup per (pair, 256-key block) and, where that still leaves CUs without
    one, per query range as well (the per-GPU share of a strongly scaled batch: 4 / 8 / 16 proteins x 512) + the reduction of
    its slabs.  Against the UNSPLIT sweep on the same inputs: dQ the same bits (the key blocks' contributions are summed in
    block order, as read-add-store did), dK / dV to rounding (query ranges are summed instead of accumulated in one chain),
    the row scales and their minimum those of the values that were written; reproducible; with and without dropout (the
    forward kernel's decisions read, or drawn again).  Against fp64 without dropout."""
    from protein_transformer_amd import kernels as K_
    D, T = 64 * H, B * L
    seq = _seq(B, L, lens, seed=11).to(dev)
    qd = rnd((T, 3 * D), 30, 1.5).to(dev)
    g = rnd((T, D), 31).to(dev)
    ar = K_.GEMM_F16X2
    res = {}
    for mode, bits in ((1, True), (2, True), (2, False)):
        kb = K_.attention_keep_bits(B, L, H, dev) if (bits and p > 0) else None
        o, lse = K_.attention_fwd(qd, seq, H, p, 77, 3, arith=ar, keep_bits=kb)
        with fused(mode):
            outs = []
            for _ in range(2):
                rs = torch.full((T,), 0x7F000000, dtype=torch.int32, device=dev)
                rm = torch.full((4,), 0x7F000000, dtype=torch.int32, device=dev)
                d = K_.attention_bwd(qd, seq, o, g, lse, H, p, 77, 3, arith=ar, row_scale=rs, row_scale_min=rm, keep_bits=kb)
                outs.append((d.clone(), rs.clone(), rm.clone()))
            torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(*outs)), (mode, bits)          # reproducible
        res[(mode, bits)] = outs[0]
    one, split, drawn = res[(1, True)], res[(2, True)], res[(2, False)]
    assert torch.equal(split[0], drawn[0])                                           # decisions read = decisions drawn
    dq1, dq2 = one[0].view(T, 3, D)[:, 0], split[0].view(T, 3, D)[:, 0]
    assert torch.equal(dq1, dq2)
    for i in (1, 2):
        a, b = one[0].view(T, 3, D)[:, i].double(), split[0].view(T, 3, D)[:, i].double()
        assert ((a - b).norm() / b.norm()).item() < 1e-6, i          # (measured 2.6e-7: another summation order, fp32)
    # row scales: the power of two that takes the largest |x| of the WRITTEN row into [2^14, 2^15); their minimum over the rows
    amax = split[0].abs().amax(1).contiguous().view(torch.int32)
    want = (torch.clamp(268 - (amax >> 23), max=254) << 23).to(torch.int32)
    assert torch.equal(split[1], want)
    assert bool((split[2] == want.min()).all())
    if p == 0.0:
        q64 = qd.double().cpu().view(B, L, 3 * D).requires_grad_()
        out, _ = ref_attention(q64, seq.cpu() != 20, H)
        out.backward(g.double().cpu().view(B, L, D))
        ref = q64.grad.view(T, 3 * D)
        assert_close(split[0], ref, 1e-4, 2e-6 * max(1.0, ref.abs().max().item()), "split sweep")


def test_split_sweep_is_the_default_for_few_pairs(dev):
    """Without PTAMD_ATTN_FUSED in the environment: head size 64 and at most half as many (protein, head) pairs as CUs take the
    split sweep (its workspace is asked for and its result is what PTAMD_ATTN_FUSED=2 gives), more pairs the unsplit one."""
    from protein_transformer_amd import _lib
    from protein_transformer_amd import kernels as K_
    assert "PTAMD_ATTN_FUSED" not in os.environ
    lib = _lib.lib()
    delta = lambda B, L, H: 4 * ((B * H * L + 3) // 4 * 4)                                           # noqa: E731
    assert lib.ptamd_attention_workspace_bytes(32, 512, 8, 64) == delta(32, 512, 8)                   # 256 pairs: unsplit
    assert lib.ptamd_attention_workspace_bytes(16, 512, 8, 64) == delta(16, 512, 8) + 4 * 2 * 8192 * 512            # key blocks only
    assert lib.ptamd_attention_workspace_bytes(4, 512, 8, 64) == delta(4, 512, 8) + 4 * (2 + 4 * 2) * 2048 * 512    # + 4 query ranges
    assert lib.ptamd_attention_workspace_bytes(4, 512, 8, 32) == delta(4, 512, 8)                     # head size 32: two kernels
    B, L, H = 4, 512, 8
    D, T = 64 * H, B * L
    seq = _seq(B, L, [512, 300, 512, 77], seed=2).to(dev)
    qd, g = rnd((T, 3 * D), 40, 1.5).to(dev), rnd((T, D), 41).to(dev)
    o, lse = K_.attention_fwd(qd, seq, H, 0.0, 0, 0, arith=K_.GEMM_AUTO)
    d_default = K_.attention_bwd(qd, seq, o, g, lse, H, 0.0, 0, 0, arith=K_.GEMM_AUTO)
    with fused(2):
        d_split = K_.attention_bwd(qd, seq, o, g, lse, H, 0.0, 0, 0, arith=K_.GEMM_AUTO)
    assert torch.equal(d_default, d_split)


def test_fused_randomised(dev):
    from protein_transformer_amd import kernels as K_
    rng = np.random.default_rng(77)
    with fused(True):
        for it in range(16):
            B, H, L = int(rng.integers(1, 4)), int(rng.choice([1, 2, 4])), int(rng.integers(2, 800))
            D = H * 64
            seq = torch.full((B, L), 20, dtype=torch.int64)
            for b in range(B):
                n = int(rng.integers(1, L + 1)) if b else L
                seq[b, :n] = torch.tensor(rng.integers(0, 20, n))
            qkv = torch.tensor(rng.normal(0, 1.2, (B, L, 3 * D)), dtype=torch.float32).double().requires_grad_()
            out, _ = ref_attention(qkv, seq != 20, H)
            dout = torch.tensor(rng.normal(0, 1, (B, L, D)), dtype=torch.float64)
            out.backward(dout)
            qd = qkv.detach().float().view(B * L, 3 * D).to(dev)
            o, lse = K_.attention_fwd(qd, seq.to(dev), H, 0.0, 0, 0, arith=K_.GEMM_F16X2)
            dqkv = K_.attention_bwd(qd, seq.to(dev), o, dout.float().view(B * L, D).to(dev), lse, H, 0.0, 0, 0, arith=K_.GEMM_F16X2)
            ref = qkv.grad.view(B * L, 3 * D)
            what = f"B={B} H={H} L={L} lens={(seq != 20).sum(1).tolist()}"
            assert_close(dqkv, ref, 1e-4, 2e-6 * max(1.0, ref.abs().max().item()), "fused attention bwd " + what)


def test_fused_dropout_masks_match_the_forward_kernel(dev):
    """Forward with V = I recovers the dropout mask; the fused backward must have drawn the same one (dense fp64 math with
    that mask), on more than one key block."""
    from protein_transformer_amd import kernels as K_
    B, L, H, p, seed, sid, dk = 2, 320, 2, 0.25, 4242, 5, 64
    D = H * dk
    seq = _seq(B, L, [L, L - 37], seed=3)
    qkv = rnd((B, L, 3 * D), 22, 1.2)
    # the mask of key block j from a forward pass whose V has the identity in rows 64 j .. 64 j + 63
    keep = torch.zeros(B, H, L, L, dtype=torch.bool)
    _, pr = ref_attention(qkv.double(), seq != 20, H)
    for j in range(L // dk):
        eye = qkv.clone()
        v = torch.zeros(L, dk)
        v[dk * j:dk * (j + 1)] = torch.eye(dk)
        eye[:, :, 2 * D:] = v[None].repeat(B, 1, H)
        pd, _ = K_.attention_fwd(eye.view(B * L, 3 * D).to(dev), seq.to(dev), H, p, seed, sid, arith=K_.GEMM_F16X2)
        keep[:, :, :, dk * j:dk * (j + 1)] = pd.view(B, L, H, dk).permute(0, 2, 1, 3).cpu() != 0
    q64 = qkv.double().requires_grad_()
    out, _ = ref_attention(q64, seq != 20, H, mask_keep=keep.double(), p=p)
    dout = rnd((B, L, D), 23).double()
    out.backward(dout)
    qd = qkv.view(B * L, 3 * D).to(dev)
    o, lse = K_.attention_fwd(qd, seq.to(dev), H, p, seed, sid, arith=K_.GEMM_F16X2)
    assert_close(o.view(B, L, D), out, 1e-5, 5e-6, "attention fwd (dropout)")
    with fused(True):
        dqkv = K_.attention_bwd(qd, seq.to(dev), o, dout.float().view(B * L, D).to(dev), lse, H, p, seed, sid, arith=K_.GEMM_F16X2)
    ref = q64.grad.view(B * L, 3 * D)
    assert_close(dqkv, ref, 1e-4, 5e-6 * max(1.0, ref.abs().max().item()), "fused attention bwd (dropout)")


@pytest.mark.parametrize("B,L,H,lens,parts", [(2, 320, 2, [320, 283], None), (1, 700, 2, [651], None), (3, 96, 8, [96, 33, 1], None),
                                              (32, 512, 8, None, None), (2, 130, 1, [130, 5], None)])
def test_keep_bits_handed_from_forward_to_fused_backward(dev, B, L, H, lens, parts):
    """ptamd_attention_fwd exports its dropout decisions (word [(b, h)][q / 32][key], bit q % 32); the fused backward kernel
    reading them gives the SAME BITS as the fused kernel drawing the generator again, the forward output does not depend on
    the export, and the exported words are the mask a forward pass with V = I shows."""
    from protein_transformer_amd import kernels as K_
    p, seed, sid, dk = 0.25, 991, 3, 64
    D = H * dk
    seq = _seq(B, L, lens if lens is not None else [L] * B, seed=5).to(dev)
    qd = rnd((B * L, 3 * D), 31, 1.2).to(dev)
    dout = rnd((B * L, D), 32).to(dev)
    bits = K_.attention_keep_bits(B, L, H, dev)
    bits.fill_(0x5A5A5A5A)
    o0, lse0 = K_.attention_fwd(qd, seq, H, p, seed, sid, arith=K_.GEMM_F16X2)
    o1, lse1 = K_.attention_fwd(qd, seq, H, p, seed, sid, arith=K_.GEMM_F16X2, keep_bits=bits)
    assert torch.equal(o0, o1) and torch.equal(lse0, lse1)
    with fused(True):
        a = K_.attention_bwd(qd, seq, o1, dout, lse1, H, p, seed, sid, arith=K_.GEMM_F16X2)
        b = K_.attention_bwd(qd, seq, o1, dout, lse1, H, p, seed, sid, arith=K_.GEMM_F16X2, keep_bits=bits)
    assert torch.equal(a, b)
    # a corrupted word must change the result (the kernel really reads them)
    bad = bits.clone()
    bad[: bad.numel() // 2] = 0
    with fused(True):
        c = K_.attention_bwd(qd, seq, o1, dout, lse1, H, p, seed, sid, arith=K_.GEMM_F16X2, keep_bits=bad)
    assert not torch.equal(a, c)
    if B * L <= 2048:   # the words against the mask a forward pass with V = I shows (first 64 keys)
        eye = qd.clone().view(B, L, 3 * D)
        v = torch.zeros(L, dk)
        v[:min(dk, L)] = torch.eye(dk)[:min(dk, L)]
        eye[:, :, 2 * D:] = v[None].repeat(B, 1, H).to(dev)
        pd, _ = K_.attention_fwd(eye.view(B * L, 3 * D), seq, H, p, seed, sid, arith=K_.GEMM_F16X2)
        shown = (pd.view(B, L, H, dk).permute(0, 2, 1, 3) != 0).cpu()            # [B, H, q, key < 64]: kept AND p > 0
        nt = (L + 31) // 32
        w = bits.view(B, H, nt, nt * 32).cpu().numpy().astype(np.uint32)
        q = np.arange(L)
        kept = ((w[:, :, q // 32, :] >> (q % 32)[None, None, :, None].astype(np.uint32)) & 1).astype(bool)   # [B, H, q, key]
        valid = (seq != 20).cpu().numpy()
        for bb in range(B):
            n = int(valid[bb].sum())
            kk = min(dk, n)
            got = kept[bb, :, :n, :kk]
            want = shown[bb, :, :n, :kk].numpy()
            # (a kept probability that underflowed to 0 would read as dropped: none at these magnitudes)
            assert (got == want).all(), f"protein {bb}: {int((got != want).sum())} decisions differ"
        frac = kept[:, :, :, :L].mean()
        assert abs(frac - (1 - p)) < 0.01


def _attn_decisions_restated(B, L, H, p, seed, sid):
    """numpy restatement of csrc/attn_dropout.h (pt_mix32 of common.h on (query, key pair) words, 16-bit halves against a
    16-bit threshold) in the layout of ptamd_attention_fwd's keep_bits: word [(b, h)][q / 32][key], bit q % 32."""
    M = np.uint64(0xFFFFFFFF)
    u = lambda x: (x & M).astype(np.uint64)                          # noqa: E731
    sh = lambda n: np.uint64(n)                                      # noqa: E731

    def mix32(x):
        x = u(x)
        x = u((~x & M) + (x << sh(15)))
        x = x ^ (x >> sh(12))
        x = u(x + (x << sh(2)))
        x = x ^ (x >> sh(4))
        x = u(x + (x << sh(3)) + (x << sh(11)))
        return u(x ^ (x >> sh(16)))
    nt = (L + 31) // 32
    lk = nt * 32
    out = np.zeros((B * H, nt, lk), dtype=np.uint32)
    thr = int(p * 65536.0 + 0.5)
    idx = np.arange(lk, dtype=np.uint64)
    for bh in range(B * H):
        lo = np.uint64((seed & 0xFFFFFFFF) ^ ((bh * 0xC2B2AE35) & 0xFFFFFFFF))
        hi = np.uint64(((seed >> 32) & 0xFFFFFFFF) ^ ((sid * 0x27D4EB2F) & 0xFFFFFFFF) ^ bh)
        qp = u(idx * np.uint64(0x9E3779B1) + lo)
        kp = u((idx >> sh(1)) * np.uint64(0x85EBCA77) + hi)
        w = mix32(qp[:, None] ^ kp[None, :])                                                    # [q, key]
        keep = ((w >> (sh(16) * (idx & sh(1)))[None, :]) & np.uint64(0xFFFF)) >= thr
        for t in range(nt):
            out[bh, t] = (keep[32 * t:32 * t + 32].astype(np.uint64) << np.arange(32, dtype=np.uint64)[:, None]).sum(0).astype(np.uint32)
    return out


@pytest.mark.parametrize("B,L,H", [(1, 64, 1), (2, 320, 2), (3, 96, 8), (2, 100, 4), (8, 512, 8)])
def test_keep_bits_are_the_generators_decisions(dev, B, L, H):
    """Every exported word against the numpy restatement of the generator - including the query tiles past the end of a
    workgroup (nothing may be written outside the buffer: a guard band behind it stays untouched)."""
    from protein_transformer_amd import kernels as K_
    p, seed, sid = 0.25, (77 << 32) | 991, 3
    D = 64 * H
    qd = rnd((B * L, 3 * D), 41).to(dev)
    seq = _seq(B, L, [L] * B).to(dev)
    n = K_.lib().ptamd_attention_keep_bits_bytes(B, L, H) // 4
    buf = torch.full((n + 4096,), 0x13572468, dtype=torch.int32, device=dev)
    K_.attention_fwd(qd, seq, H, p, seed, sid, arith=K_.GEMM_F16X2, keep_bits=buf[:n])
    nt = (L + 31) // 32
    got = buf[:n].cpu().numpy().view(np.uint32).reshape(B * H, nt, nt * 32)
    want = _attn_decisions_restated(B, L, H, p, seed, sid)
    qmask = np.array([sum(1 << i for i in range(32) if 32 * t + i < L) for t in range(nt)], dtype=np.uint32)   # queries < L
    diff = (got ^ want) & qmask[None, :, None]
    assert not diff[:, :, :L].any(), f"{int(np.count_nonzero(diff[:, :, :L]))} words differ"
    assert (buf[n:] == 0x13572468).all()


def test_keep_bits_only_in_the_f16x2_arithmetic(dev):
    from protein_transformer_amd import kernels as K_
    B, L, H = 1, 64, 2
    qd = rnd((B * L, 3 * 64 * H), 33).to(dev)
    seq = _seq(B, L, [L]).to(dev)
    bits = K_.attention_keep_bits(B, L, H, dev)
    assert bits.numel() == B * H * 2 * 64
    with pytest.raises(RuntimeError):
        K_.attention_fwd(qd, seq, H, 0.1, 1, 1, arith=K_.GEMM_BF16X3, keep_bits=bits)
    assert K_.attention_bwd_reads_keep_bits(32, 512, 8, 64, K_.GEMM_AUTO)
    assert not K_.attention_bwd_reads_keep_bits(32, 512, 8, 64, K_.GEMM_BF16X3)
    assert K_.attention_bwd_reads_keep_bits(4, 512, 16, 32, K_.GEMM_AUTO)          # the dK / dV kernel of the two-kernel path
    assert not K_.attention_bwd_reads_keep_bits(4, 512, 64, 8, K_.GEMM_AUTO)       # (head size 8: the exact-f32 kernels)


@pytest.mark.parametrize("B,L,H,dk,lens", [(2, 320, 2, 64, [320, 283]), (4, 512, 8, 64, None), (3, 96, 8, 64, [96, 33, 1]),
                                           (2, 256, 8, 32, [256, 200]), (16, 256, 8, 32, None), (1, 130, 1, 32, [101]),
                                           (8, 512, 8, 64, None)])
def test_keep_bits_on_the_two_kernel_path(dev, B, L, H, dk, lens):
    """Small batches take dQ and dK / dV from two kernels; the dK / dV kernel (keys in lanes, like the one-sweep kernel) reads
    the forward kernel's decisions, the dQ kernel draws them: same bits as both drawing them, for both head sizes and every
    workgroup shape the launcher picks."""
    from protein_transformer_amd import kernels as K_
    p, seed, sid = 0.2, 4711, 6
    D = H * dk
    seq = _seq(B, L, lens if lens is not None else [L] * B, seed=9).to(dev)
    qd = rnd((B * L, 3 * D), 51, 1.1).to(dev)
    dout = rnd((B * L, D), 52).to(dev)
    bits = K_.attention_keep_bits(B, L, H, dev)
    bits.fill_(0x0F0F0F0F)
    o, lse = K_.attention_fwd(qd, seq, H, p, seed, sid, arith=K_.GEMM_F16X2, keep_bits=bits)
    with fused(False):
        a = K_.attention_bwd(qd, seq, o, dout, lse, H, p, seed, sid, arith=K_.GEMM_F16X2)
        b = K_.attention_bwd(qd, seq, o, dout, lse, H, p, seed, sid, arith=K_.GEMM_F16X2, keep_bits=bits)
        bad = bits.clone()
        bad[: bad.numel() // 2] = 0
        c = K_.attention_bwd(qd, seq, o, dout, lse, H, p, seed, sid, arith=K_.GEMM_F16X2, keep_bits=bad)
    assert torch.equal(a, b)
    assert not torch.equal(a, c)          # the kernel really reads them



def test_fused_wide_row_ranges_and_degenerate_rows(dev):
    from protein_transformer_amd import kernels as K_
    B, L, H, dk = 2, 512, 8, 64
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
    with fused(True):
        dq = K_.attention_bwd(qd, seq.to(dev), o, dout.float().view(B * L, D).to(dev), lse, H, 0.0, 0, 0, arith=K_.GEMM_F16X2)
    assert torch.isfinite(dq).all()
    gr = qkv.grad
    for i, name in enumerate(("dQ", "dK", "dV")):
        a, r = dq.view(B, L, 3 * D)[:, :, i * D:(i + 1) * D].double().cpu(), gr[:, :, i * D:(i + 1) * D]
        assert ((a - r).norm() / r.norm()).item() < 2e-6, name
    # degenerate rows: zero K / V groups and tiles, a fully padded tile, huge and tiny magnitudes, zero gradients
    B, L, H = 2, 300, 2
    D = H * dk
    qkv = torch.randn(B, L, 3 * D, generator=g)
    qkv[0, 32:72, D:] = 0.0
    qkv[1, :, 2 * D:] *= 3000.0
    qkv[1, 5:9, 2 * D:] *= 1e-12
    seq = torch.randint(0, 20, (B, L), generator=g)
    seq[1, 96:] = 20
    dout = torch.randn(B, L, D, generator=g) * 1e-20
    qd = qkv.view(B * L, 3 * D).to(dev)
    res = {}
    for mode, on in ((K_.GEMM_F32, False), (K_.GEMM_F16X2, True)):
        o, lse = K_.attention_fwd(qd, seq.to(dev), H, 0.0, 0, 0, arith=mode)
        with fused(on):
            dq = K_.attention_bwd(qd, seq.to(dev), o, dout.view(B * L, D).to(dev), lse, H, 0.0, 0, 0, arith=mode)
        assert torch.isfinite(dq).all()
        res[mode] = dq.double().cpu()
    assert ((res[K_.GEMM_F32] - res[K_.GEMM_F16X2]).norm() / res[K_.GEMM_F32].norm()).item() < 3e-6
    with fused(True):
        for dz in (torch.zeros(B, L, D), torch.cat([torch.zeros(1, L, D), torch.randn(1, L, D, generator=g)])):
            o, lse = K_.attention_fwd(qd, seq.to(dev), H, 0.1, 3, 1, arith=K_.GEMM_F16X2)
            dq = K_.attention_bwd(qd, seq.to(dev), o, dz.view(B * L, D).to(dev), lse, H, 0.1, 3, 1, arith=K_.GEMM_F16X2)
            assert torch.isfinite(dq).all() and (dq.view(B, L, 3 * D)[0] == 0).all()


def test_fused_row_scales_and_automatic_choice(dev):
    """The f16x2 row scales of dqkv the fused kernel leaves behind are those of a pass over what it wrote; with 17 proteins
    x 8 heads = 136 workgroups (more than half of the 256 CUs) the library takes the fused kernel by itself."""
    from test_gpu_scales import as_float, scale_of
    from protein_transformer_amd import kernels as K
    B, L, H, dk = 17, 300, 8, 64
    g = torch.Generator().manual_seed(3)
    D = H * dk
    qkv = torch.randn(B * L, 3 * D, generator=g).to(dev)
    dout = (torch.randn(B * L, D, generator=g) * torch.exp(2 * torch.randn(B * L, 1, generator=g))).to(dev) * 1e-3
    seq = torch.randint(0, 20, (B, L), generator=g)
    seq[0, L - 40:] = 20
    seq = seq.to(dev)
    o, lse = K.attention_fwd(qkv, seq, H, 0.1, 5, 2, arith=K.GEMM_F16X2)
    out = {}
    for name, ctx in (("auto", None), ("fused", fused(True)), ("two", fused(False))):
        rs = torch.full((B * L,), 0x7F000000, dtype=torch.int32, device=dev)
        mn = torch.full((4,), 0x7F000000, dtype=torch.int32, device=dev)
        if ctx is None:
            assert "PTAMD_ATTN_FUSED" not in os.environ
            dq = K.attention_bwd(qkv, seq, o, dout, lse, H, 0.1, 5, 2, arith=K.GEMM_F16X2, row_scale=rs, row_scale_min=mn)
        else:
            with ctx:
                dq = K.attention_bwd(qkv, seq, o, dout, lse, H, 0.1, 5, 2, arith=K.GEMM_F16X2, row_scale=rs, row_scale_min=mn)
        want = scale_of(dq.abs().amax(dim=1).cpu().numpy())
        assert np.array_equal(as_float(rs), want), name
        assert np.array_equal(as_float(mn), np.full(4, want.min())), name
        out[name] = dq
    assert torch.equal(out["auto"], out["fused"])                    # 136 workgroups: the library chose the fused kernel
    assert not torch.equal(out["fused"], out["two"])                 # (another summation order of dQ)
    assert ((out["fused"] - out["two"]).norm() / out["two"].norm()).item() < 1e-6
