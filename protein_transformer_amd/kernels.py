"""Thin Python entry points for the encoder / optimizer kernels of libptamd (include/ptamd.h).

Plumbing only: tensors come from the PyTorch-ROCm caching allocator, kernels are enqueued on the
current HIP stream through the C ABI.  Everything here requires device tensors; nothing falls back.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import GemmArgs, GemmHpArgs, check, lib, ptr, stream, workspace

EPI_RELU, EPI_TANH, EPI_ACCUM, EPI_GATE, EPI_SLABS = 1, 2, 4, 8, 16

# bench.py sets this to a list to time every GEMM launch with HIP events recorded on the launch stream:
# entries are (flops, start_event, end_event).  None (the default) adds no work to the hot path.
GEMM_TIMING = None
GEMM_EVENT_POOL = []
GEMM_BYTES = []          # algorithmic operand bytes (A + B + C [+ residual, + accumulate]) of every timed launch
GEMM_RESERVED_CUS = 0    # CUs the persistent GEMMs leave free (dp.reserve_cus_for_collectives sets it under data parallelism)


def gemm(A, B, C_out, *, M, N, K, lda, ldb, ldc, a_kmajor=False, b_kmajor=False, bias=None, residual=None,
         ldr=0, flags=0, dropout_p=0.0, seed=0, stream_id=0, split_k=1, colsum=None, gate_scale=0.0, arith=None,
         a_scale=None, a_scale_stride=1, b_scale=None, b_scale_stride=1, gate_mask=None, ws=None):
    """C[M,N] = epilogue(A (*) B); operand layouts as documented in ptamd.h.  `arith`: GEMM_* constant of this call
    (None = the host-side default, `set_gemm_mode`); the library itself keeps no mode.  `ws`: the caller's own workspace
    (EPI_SLABS: the K slices stay in it; C_out may then be None)."""
    # split-K slabs and, for the f16x2 arithmetic, the row scales of the two operands
    if ws is None:
        ws = workspace("gemm", lib().ptamd_gemm_workspace_bytes(M, N, split_k), A.device)
    args = GemmArgs(M=M, N=N, K=K, A=A.data_ptr(), lda=lda, a_kmajor=int(a_kmajor), B=B.data_ptr(), ldb=ldb,
                    b_kmajor=int(b_kmajor), C=C_out.data_ptr() if C_out is not None else ws.data_ptr(), ldc=ldc,
                    bias=bias.data_ptr() if bias is not None else None,
                    residual=residual.data_ptr() if residual is not None else None, ldr=ldr, flags=flags,
                    dropout_p=float(dropout_p), seed=int(seed) & (2 ** 64 - 1), stream_id=int(stream_id),
                    split_k=int(split_k), workspace=ws.data_ptr() if ws is not None else None,
                    workspace_bytes=ws.numel() if ws is not None else 0,
                    colsum=colsum.data_ptr() if colsum is not None else None, gate_scale=float(gate_scale),
                    arith=int(_DEFAULT_ARITH if arith is None else arith), reserved_cus=int(GEMM_RESERVED_CUS),
                    a_scale=a_scale.data_ptr() if a_scale is not None else None, a_scale_stride=int(a_scale_stride),
                    b_scale=b_scale.data_ptr() if b_scale is not None else None, b_scale_stride=int(b_scale_stride),
                    gate_mask=gate_mask.data_ptr() if gate_mask is not None else None)
    if GEMM_TIMING is None:
        check(lib().ptamd_gemm(C.byref(args), stream()), "gemm")
    else:
        # events come from a pool created before the timed region: the hot loop only pays two hipEventRecord calls
        e0, e1 = GEMM_EVENT_POOL.pop(), GEMM_EVENT_POOL.pop()
        e0.record()
        check(lib().ptamd_gemm(C.byref(args), stream()), "gemm")
        e1.record()
        # (flops, events, matrix-pipe products per fp32 product of the arithmetic this call ran in)
        # (flops, events, matrix-pipe products per fp32 product of the arithmetic this call ran in, which kernel)
        nprod = lib().ptamd_gemm_products(C.byref(args))
        kind = "dW" if (a_kmajor and b_kmajor) else ("dX" if b_kmajor else "fwd")
        GEMM_TIMING.append((2.0 * M * N * K, e0, e1, nprod,
                            f"ptamd_gemm {kind} (staging kernel, A_KMAJOR={int(bool(a_kmajor))} B_KMAJOR={int(bool(b_kmajor))} NPROD={nprod})"))
        GEMM_BYTES.append(4 * (M * K + N * K + M * N * (1 + (residual is not None) + bool(flags & EPI_ACCUM))))
    return C_out


GEMM_F32, GEMM_BF16X3, GEMM_BF16X3_FULL, GEMM_F16X2, GEMM_AUTO = 0, 1, 2, 3, 4   # ptamd.h: PTAMD_GEMM_*




def _env_arith():
    import os
    try:
        v = int(os.environ.get("PTAMD_GEMM_MODE", GEMM_AUTO))
    except ValueError:
        v = GEMM_AUTO
    return v if v in (GEMM_F32, GEMM_BF16X3, GEMM_BF16X3_FULL, GEMM_F16X2, GEMM_AUTO) else GEMM_AUTO


# Host-side DEFAULT arithmetic of calls / models that do not name one (PTAMD_GEMM_* of include/ptamd.h).  The library has
# no mode of its own: every ptamd_gemm / ptamd_attention call carries its arithmetic, and a model keeps the one it was
# given (`model.gemm_mode`), so two models with different arithmetics can run side by side in one process.
_DEFAULT_ARITH = _env_arith()


def set_gemm_mode(mode):
    """Set the default arithmetic for calls and models that do not carry their own (`arith=` / `model.gemm_mode`)."""
    global _DEFAULT_ARITH
    if int(mode) not in (GEMM_F32, GEMM_BF16X3, GEMM_BF16X3_FULL, GEMM_F16X2, GEMM_AUTO):
        raise ValueError(f"unknown GEMM arithmetic {mode}")
    _DEFAULT_ARITH = int(mode)


def get_gemm_mode():
    return _DEFAULT_ARITH


class HpOperand:
    """An fp32 matrix [rows, K] in the pre-split "half-pair" format of include/ptamd.h (two f16 planes in MFMA-operand
    blocks + one power-of-two scale per row): what ptamd_gemm_hp reads by LDS-DMA."""
    __slots__ = ("planes", "scale", "rows", "K")

    def __init__(self, rows, K, device):
        self.rows, self.K = int(rows), int(K)
        self.planes = torch.empty(lib().ptamd_hp_bytes(self.rows, self.K), dtype=torch.uint8, device=device)
        self.scale = torch.empty(lib().ptamd_hp_padded_rows(self.rows), dtype=torch.float32, device=device)


def hp_split(x, transposed=False, out=None):
    """fp32 [rows, K] (or, transposed, [K, rows]) -> HpOperand."""
    assert x.dim() == 2 and x.stride(1) == 1
    rows, K = (x.shape[1], x.shape[0]) if transposed else (x.shape[0], x.shape[1])
    if out is None:
        out = HpOperand(rows, K, x.device)
    check(lib().ptamd_hp_split(ptr(x), x.stride(0), rows, K, int(transposed), ptr(out.planes), ptr(out.scale), stream()),
          "hp_split")
    return out


def gemm_hp(a, b, C_out, *, bias=None, residual=None, ldr=0, flags=0, dropout_p=0.0, seed=0, stream_id=0, split_k=1,
            gate_scale=0.0, gate_mask=None, gate_mask_out=None, kv=None, kv_col0=0, kv_heads=0):
    """C[M,N] = epilogue(A B^T) from pre-split operands a = hp [M,K], b = hp [N,K].  `kv` = (planes, inverse scales) of
    `attention_kv_buffers`: the columns from kv_col0 on (K | V of the QKV product, kv_heads heads of 64) leave as the pre-split
    planes the f16x2 attention kernels read instead of fp32 (C[:, kv_col0:] is then not written)."""
    assert a.K == b.K
    M, N = a.rows, b.rows
    ws = workspace("gemm_hp", lib().ptamd_gemm_hp_workspace_bytes(M, N, split_k), C_out.device) if split_k > 1 else None
    args = GemmHpArgs(M=M, N=N, K=a.K, A=a.planes.data_ptr(), A_scale=a.scale.data_ptr(), B=b.planes.data_ptr(),
                      B_scale=b.scale.data_ptr(), C=C_out.data_ptr(), ldc=C_out.stride(0),
                      bias=bias.data_ptr() if bias is not None else None,
                      residual=residual.data_ptr() if residual is not None else None, ldr=ldr, flags=flags,
                      dropout_p=float(dropout_p), seed=int(seed) & (2 ** 64 - 1), stream_id=int(stream_id),
                      split_k=int(split_k), workspace=ws.data_ptr() if ws is not None else None,
                      workspace_bytes=ws.numel() if ws is not None else 0, gate_scale=float(gate_scale),
                      reserved_cus=int(GEMM_RESERVED_CUS),
                      gate_mask=gate_mask.data_ptr() if gate_mask is not None else None,
                      gate_mask_out=gate_mask_out.data_ptr() if gate_mask_out is not None else None,
                      kv_planes=kv[0].data_ptr() if kv is not None else None, kv_inv=kv[1].data_ptr() if kv is not None else None,
                      kv_col0=int(kv_col0), kv_heads=int(kv_heads))
    if GEMM_TIMING is None:
        check(lib().ptamd_gemm_hp(C.byref(args), stream()), "gemm_hp")
    else:
        e0, e1 = GEMM_EVENT_POOL.pop(), GEMM_EVENT_POOL.pop()
        e0.record()
        check(lib().ptamd_gemm_hp(C.byref(args), stream()), "gemm_hp")
        e1.record()
        GEMM_TIMING.append((2.0 * M * N * a.K, e0, e1, 3, "ptamd_gemm_hp (LDS-DMA kernel gemm_hp3_kernel, NPROD=3)"))
        GEMM_BYTES.append(4 * (M * a.K + N * a.K + M * N * (1 + (residual is not None) + bool(flags & EPI_ACCUM))))
    return C_out


DW_SLOTS = 512           # knob of pick_split_k (A/B of round 3: profiles/r03, script in git history 160951b)


def pick_split_k(M, N, K, slots=None):
    """K splits for a reduction-heavy product with few output tiles: as many (tile, split) items as fit in ONE round
    of the persistent grid (2 workgroups x 256 CUs) - one item more than that would cost a whole second round."""
    slots = DW_SLOTS if slots is None else slots
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles >= slots // 2 or K <= 512:
        return 1
    return max(1, min(K // 256, slots // tiles))


SPLIT_K_ROWS = True      # knob of pick_split_k_rows (tests compare both settings)


def pick_split_k_rows(M, N, K, slots=256):
    """K splits for an activation x weight product whose output tiles (128 x 128: the staging GEMM halves its 256-row tile
    when that finishes sooner) do not fill the chip (few tokens: small batches, the per-GPU share of a strongly scaled
    batch) and whose reduction is long enough to be worth cutting: the slabs are summed in a fixed order by the reduce
    kernel, which also applies the epilogue (same dropout masks)."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if not SPLIT_K_ROWS or tiles * 2 > slots or K < 1024:
        return 1
    return max(1, min(K // 512, slots // tiles))


def linear_fwd(x, w, b, out=None, defer_reduce=False, **epi):
    """y[T,N] = x[T,K] w[N,K]^T + b with a fused epilogue.  `defer_reduce` (the caller hands the result to layernorm_fwd
    and nothing else): where the product is split over K (few tokens) and its epilogue is bias + dropout + residual, the K
    slices stay unreduced and a `PendingRows` is returned - that LayerNorm makes the rows as it reads them (same bits)."""
    T, K = x.shape
    N = w.shape[0]
    sk = pick_split_k_rows(T, N, K)
    if defer_reduce and DEFER_REDUCE and sk > 1 and out is None and b is not None and N % 4 == 0 and N <= 512 and \
            epi.get("flags", 0) == 0 and epi.get("residual") is not None and epi.get("ldr") == N and \
            epi["residual"].is_contiguous() and int(_DEFAULT_ARITH if epi.get("arith") is None else epi["arith"]) != GEMM_F32:
        n = effective_splits(K, sk)
        if 2 <= n <= 4:
            ws = workspace("gemm_slabs", lib().ptamd_gemm_workspace_bytes(T, N, sk), x.device)
            rest = {k: v for k, v in epi.items() if k not in ("residual", "ldr", "dropout_p", "seed", "stream_id", "flags")}
            gemm(x, w, None, M=T, N=N, K=K, lda=x.stride(0), ldb=w.stride(0), ldc=N, split_k=sk, flags=EPI_SLABS, ws=ws, **rest)
            return PendingRows(Slabs(ws, n, T * N, (T, N)), b, epi["residual"], epi.get("dropout_p", 0.0), epi.get("seed", 0),
                               epi.get("stream_id", 0))
    if out is None:
        out = torch.empty(T, N, dtype=torch.float32, device=x.device)
    return gemm(x, w, out, M=T, N=N, K=K, lda=x.stride(0), ldb=w.stride(0), ldc=out.stride(0), bias=b,
                split_k=sk, **epi)


def gate_mask_buffer(M, N, device):
    """Buffer for the 1-bit gate of an [M, N] activation (ptamd_gemm_hp gate_mask_out -> ptamd_gemm gate_mask)."""
    return torch.empty(lib().ptamd_gate_mask_bytes(M, N) // 8, dtype=torch.int64, device=device)


class Slabs:
    """The K slices of a split product left unreduced (EPI_SLABS): `n` [T, D] fp32 slabs `stride` floats apart in `buf`;
    their sum in slab order is the product.  Read by layernorm_bwd_dropout (dy)."""
    __slots__ = ("buf", "n", "stride", "shape")

    def __init__(self, buf, n, stride, shape):
        self.buf, self.n, self.stride, self.shape = buf, n, stride, shape

    def data_ptr(self):
        return self.buf.data_ptr()

    def sum(self):     # (tests: the reduction launch's order)
        T, D = self.shape
        v = self.buf[:self.n * self.stride * 4].view(torch.float32).view(self.n, self.stride)[:, :T * D]
        out = v[0].clone()
        for k in range(1, self.n):
            out += v[k]
        return out.view(T, D)


class PendingRows:
    """x = residual + dropout(product + bias) whose product is still K slices (`slabs`): made by the LayerNorm forward that
    reads it (layernorm_fwd -> ptamd_layernorm_fwd_sum), which leaves the rows in `.value`."""
    __slots__ = ("slabs", "bias", "residual", "dropout_p", "seed", "stream_id", "value")

    def __init__(self, slabs, bias, residual, dropout_p, seed, stream_id):
        self.slabs, self.bias, self.residual = slabs, bias, residual
        self.dropout_p, self.seed, self.stream_id, self.value = float(dropout_p), int(seed) & (2 ** 64 - 1), int(stream_id), None

    @property
    def shape(self):
        return self.slabs.shape

    @property
    def device(self):
        return self.residual.device


DEFER_REDUCE = os.environ.get("PTAMD_DEFER_REDUCE", "1") != "0"      # knob for A/B and tests


def effective_splits(K, split_k):
    """The K slices ptamd_gemm makes of a reduction of length K asked to split `split_k` ways (whole 32-blocks each)."""
    kblocks = -(-K // 32)
    splits = max(1, min(int(split_k), kblocks))
    per = -(-kblocks // splits) * 32
    return -(-K // per)


def linear_bwd_input(dy, w, out=None, flags=0, gate=None, gate_dropout_p=0.0, arith=None, gate_mask=None, defer_reduce=False,
                     **scales):
    """dx[T,K] = dy[T,N] w[N,K];  with `gate` (the saved output of a ReLU + dropout layer, [T,K]) the product is passed
    through the backward of that layer in the epilogue: dx = gate > 0 ? dx / (1 - p) : 0.  `gate_mask`: the same gate as one
    bit per element (written by the product that made `gate`): read instead of `gate` where the kernel can (same bits).
    `defer_reduce`: where the product is split over K (few tokens) and the split is one layernorm_bwd_dropout can sum, the
    K slices are returned unreduced (a `Slabs`; no reduction launch) for that kernel to add as it reads them - same bits."""
    T, N = dy.shape
    K = w.shape[1]
    sk = pick_split_k_rows(T, K, N)
    if defer_reduce and DEFER_REDUCE and sk > 1 and out is None and gate is None and gate_mask is None and flags == 0 and \
            K % 4 == 0 and K <= 512 and int(_DEFAULT_ARITH if arith is None else arith) != GEMM_F32:
        n = effective_splits(N, sk)
        if 2 <= n <= 4:
            # the slabs' own workspace (the shared one is the next product's); the kernel that sums them is enqueued on
            # this stream before the next product of this kind writes it
            ws = workspace("gemm_slabs", lib().ptamd_gemm_workspace_bytes(T, K, sk), dy.device)
            gemm(dy, w, None, M=T, N=K, K=N, lda=dy.stride(0), ldb=w.stride(0), ldc=K, b_kmajor=True, split_k=sk,
                 flags=EPI_SLABS, arith=arith, ws=ws, **scales)
            return Slabs(ws, n, T * K, (T, K))
    if out is None:
        out = torch.empty(T, K, dtype=torch.float32, device=dy.device)
    if gate_mask is not None and sk == 1 and K % 4 == 0 and out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0 and \
            int(_DEFAULT_ARITH if arith is None else arith) in (GEMM_F16X2, GEMM_AUTO) and N >= 32 and \
            dy.shape[0] * dy.stride(0) * 4 < 2 ** 32 and w.shape[0] * w.stride(0) * 4 < 2 ** 32:   # (else ptamd_gemm runs in f32)
        return gemm(dy, w, out, M=T, N=K, K=N, lda=dy.stride(0), ldb=w.stride(0), ldc=out.stride(0), b_kmajor=True, split_k=1,
                    flags=flags | EPI_GATE, gate_mask=gate_mask, gate_scale=1.0 / (1.0 - gate_dropout_p), arith=arith, **scales)
    if gate_mask is not None and gate is None:
        # the 1-bit gate is only read by the unsplit f16x2 product with a float4 epilogue: without the fp32 activation to fall
        # back on, this call would silently return an UNGATED product
        raise ValueError("linear_bwd_input: gate_mask given, but this product cannot read it (split-K, arithmetic, alignment "
                         "or operand size) and no fp32 `gate` was passed to fall back on")
    if gate is not None:
        return gemm(dy, w, out, M=T, N=K, K=N, lda=dy.stride(0), ldb=w.stride(0), ldc=out.stride(0), b_kmajor=True, split_k=sk,
                    flags=flags | EPI_GATE, residual=gate, ldr=gate.stride(0), gate_scale=1.0 / (1.0 - gate_dropout_p),
                    arith=arith, **scales)
    return gemm(dy, w, out, M=T, N=K, K=N, lda=dy.stride(0), ldb=w.stride(0), ldc=out.stride(0), b_kmajor=True, split_k=sk,
                flags=flags, arith=arith, **scales)


def linear_bwd_weight(dy, x, dw, dbias=None, arith=None, dy_scale=None, x_scale=None):
    """dw[N,K] += dy[T,N]^T x[T,K]  (reduction over the T tokens, split across workgroups) and, fused into the same
    pass over dy, dbias[N] += sum_t dy[t,N]."""
    T, N = dy.shape
    K = x.shape[1]
    if dy_scale is not None and x_scale is not None:   # (the caller passes them only in the AUTO / F16X2 arithmetics)
        # both operands come with ONE scale each (four copies; the scale of their largest row): the product runs in f16x2
        # arithmetic - half the matrix-pipe work of bf16x3 - without a pass over the two big operands
        return gemm(dy, x, dw, M=N, N=K, K=T, lda=dy.stride(0), ldb=x.stride(0), ldc=dw.stride(0), a_kmajor=True,
                    b_kmajor=True, flags=EPI_ACCUM, split_k=pick_split_k(N, K, T), colsum=dbias, arith=GEMM_F16X2,
                    a_scale=dy_scale, a_scale_stride=0, b_scale=x_scale, b_scale_stride=0)
    return gemm(dy, x, dw, M=N, N=K, K=T, lda=dy.stride(0), ldb=x.stride(0), ldc=dw.stride(0), a_kmajor=True,
                b_kmajor=True, flags=EPI_ACCUM, split_k=pick_split_k(N, K, T), colsum=dbias, arith=arith)


GROUP_DW = True          # the weight-gradient products of a layer in one launch where they qualify (knob for A/B and tests)


GROUP_MIN_K = int(os.environ.get("PTAMD_GROUP_MIN_K", 1024))      # fewest tokens per item of a group (knob for measurements)


def pick_group_split(tiles256, T, slots=256):
    """Common K split of a GROUP of weight-gradient products (`tiles256` output tiles of 256 x 128 in all, T tokens): the
    items are dealt out in contiguous ranges of the whole list, so what counts is rounds x (tokens per item + a fixed
    cost per item: prologue, slab write and its reduction - about 640 tokens' worth).  0: not worth a group (too few tokens)."""
    best, best_cost = 0, None
    for s in range(2, max(2, T // GROUP_MIN_K) + 1):
        rounds = -(-tiles256 * s // slots)
        cost = rounds * (-(-T // s) + 640)
        if best_cost is None or cost < best_cost:
            best, best_cost = s, cost
    return best if T >= 2048 else 0


def linear_bwd_weight_group(jobs, split_k):
    """The products of `linear_bwd_weight` for several (dy, x, dw, dbias, dy_scale, x_scale) at once: ONE launch for the
    products, one for all their split-K reductions (ptamd_gemm_group; every job f16x2 with uniform scales).  Bit for bit
    what the separate calls give with the same `split_k`."""
    arr = (GemmArgs * len(jobs))()
    flops = 0.0
    nbytes = 0
    for j, (dy, x, dw, dbias, dy_scale, x_scale) in enumerate(jobs):
        T, N = dy.shape
        Kd = x.shape[1]
        ws = workspace(f"gemm_group{j}", lib().ptamd_gemm_workspace_bytes(N, Kd, split_k), dw.device)
        arr[j] = GemmArgs(M=N, N=Kd, K=T, A=dy.data_ptr(), lda=dy.stride(0), a_kmajor=1, B=x.data_ptr(), ldb=x.stride(0),
                          b_kmajor=1, C=dw.data_ptr(), ldc=dw.stride(0), bias=None, residual=None, ldr=0, flags=EPI_ACCUM,
                          dropout_p=0.0, seed=0, stream_id=0, split_k=int(split_k), workspace=ws.data_ptr(),
                          workspace_bytes=ws.numel(), colsum=dbias.data_ptr() if dbias is not None else None, gate_scale=0.0,
                          arith=GEMM_F16X2, reserved_cus=int(GEMM_RESERVED_CUS), a_scale=dy_scale.data_ptr(), a_scale_stride=0,
                          b_scale=x_scale.data_ptr(), b_scale_stride=0)
        flops += 2.0 * N * Kd * T
        nbytes += 4 * (N * T + Kd * T + 2 * N * Kd)
    if GEMM_TIMING is None:
        check(lib().ptamd_gemm_group(arr, len(jobs), stream()), "gemm_group")
    else:
        e0, e1 = GEMM_EVENT_POOL.pop(), GEMM_EVENT_POOL.pop()
        e0.record()
        check(lib().ptamd_gemm_group(arr, len(jobs), stream()), "gemm_group")
        e1.record()
        GEMM_TIMING.append((flops, e0, e1, 3, "ptamd_gemm_group dW (staging kernel on a group + one split-K reduce, NPROD=3)"))
        GEMM_BYTES.append(nbytes)


def colsum(x, out, accumulate=True):
    T, N = x.shape
    ws = workspace("colsum", lib().ptamd_colsum_workspace_bytes(N), x.device)
    check(lib().ptamd_colsum(ptr(x), T, N, x.stride(0), int(accumulate), ptr(out), ptr(ws), ws.numel(), stream()),
          "colsum")
    return out


def layernorm_fwd(x, gamma, beta, row_scale=None, planes=None):
    """y, mean, rstd; `row_scale` (uint32-as-int32 [T], optional) receives the f16x2 scale of every row of y; `planes`
    (uint8 buffer of ptamd_hp_bytes(T, D), optional, needs row_scale) receives y once more in the pre-split hp format."""
    T, D = x.shape
    if isinstance(x, PendingRows):           # (linear_fwd: defer_reduce) the rows are made here and left in x.value
        r = x.residual
        x.value, y = torch.empty_like(r), torch.empty_like(r)
        mean = torch.empty(T, dtype=torch.float32, device=r.device)
        rstd = torch.empty(T, dtype=torch.float32, device=r.device)
        check(lib().ptamd_layernorm_fwd_sum(ptr(x.slabs), x.slabs.n, x.slabs.stride, ptr(x.bias), ptr(r), x.dropout_p, x.seed,
                                            x.stream_id, ptr(x.value), ptr(gamma), ptr(beta), T, D, ptr(y), ptr(mean), ptr(rstd),
                                            ptr(row_scale), ptr(planes), stream()), "layernorm_fwd_sum")
        return y, mean, rstd
    y = torch.empty_like(x)
    mean = torch.empty(T, dtype=torch.float32, device=x.device)
    rstd = torch.empty(T, dtype=torch.float32, device=x.device)
    check(lib().ptamd_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), T, D, ptr(y), ptr(mean), ptr(rstd), ptr(row_scale),
                                    ptr(planes), stream()), "layernorm_fwd")
    return y, mean, rstd


def hp_view(planes, scale, rows, K):
    """HpOperand over existing buffers (planes: uint8, scale: float32 / int32 bits of powers of two, one per row)."""
    op = HpOperand.__new__(HpOperand)
    op.rows, op.K, op.planes = int(rows), int(K), planes
    op.scale = scale.view(torch.float32) if scale.dtype != torch.float32 else scale
    return op


def hp_split_rows(mats, outs):
    """K-contiguous matrices -> HpOperands `outs` (allocated by the caller) in one launch per 16 matrices."""
    from ._lib import HpSplitJob
    for i in range(0, len(mats), 16):
        chunk = list(zip(mats[i:i + 16], outs[i:i + 16]))
        arr = (HpSplitJob * len(chunk))()
        for k, (w, o) in enumerate(chunk):
            assert w.dim() == 2 and w.stride(1) == 1 and (o.rows, o.K) == tuple(w.shape)
            arr[k] = HpSplitJob(x=w.data_ptr(), ld=w.stride(0), rows=w.shape[0], K=w.shape[1], planes=o.planes.data_ptr(),
                                scale=o.scale.data_ptr())
        check(lib().ptamd_hp_split_rows(arr, len(chunk), stream()), "hp_split_rows")
    return outs


def hp_split_cols(mats, scales, outs):
    """Row-contiguous matrices [K, rows] -> HpOperands of their transposes (rows x K) with GIVEN row scales (int32 / float32
    tensors holding powers of two, e.g. the column scales of ptamd_weight_scales) in one launch per 16 matrices."""
    from ._lib import HpSplitJob
    for i in range(0, len(mats), 16):
        chunk = list(zip(mats[i:i + 16], scales[i:i + 16], outs[i:i + 16]))
        arr = (HpSplitJob * len(chunk))()
        for k, (w, sc, o) in enumerate(chunk):
            assert w.dim() == 2 and w.stride(1) == 1 and (o.rows, o.K) == (w.shape[1], w.shape[0]) and sc.numel() >= o.rows
            arr[k] = HpSplitJob(x=w.data_ptr(), ld=w.stride(0), rows=o.rows, K=o.K, planes=o.planes.data_ptr(), scale=sc.data_ptr())
        check(lib().ptamd_hp_split_cols(arr, len(chunk), stream()), "hp_split_cols")
    return outs


def _ln_workspace(D, device, pending):
    """Workspace of one LayerNorm backward; with a `pending` list (deferred reduction) every pending site gets its own."""
    tag = "ln" if pending is None else f"ln{len(pending)}"
    return workspace(tag, lib().ptamd_layernorm_bwd_workspace_bytes(D), device)


def layernorm_bwd_flush(pending):
    """Finish the deferred (dgamma, dbeta) reductions in `pending` with one launch per 16 sites; empties the list."""
    from ._lib import LnReduceJob
    for i in range(0, len(pending), 16):
        chunk = pending[i:i + 16]
        arr = (LnReduceJob * len(chunk))()
        for k, (ws, D, dg, db) in enumerate(chunk):
            arr[k] = LnReduceJob(partials=ws.data_ptr(), D=D, dgamma=dg.data_ptr(), dbeta=db.data_ptr())
        check(lib().ptamd_layernorm_bwd_reduce(arr, len(chunk), stream()), "layernorm_bwd_reduce")
    del pending[:]


def layernorm_bwd_dropout(dy, x, gamma, mean, rstd, dgamma, dbeta, dres, dropout_p, seed, stream_id, row_scale=None,
                          bound_factor=None, bound_scale=None, row_scale_min=None, bound_scale_min=None, pending=None,
                          planes=None):
    """LayerNorm backward fused with the dropout backward of its output (ptamd_layernorm_bwd_dropout): returns
    (dx, dropped); dropped is dx itself when dropout_p == 0.  Fills row_scale / bound_scale [T] when given; `planes`
    (uint8 buffer of ptamd_hp_bytes(T, D), needs row_scale) receives `dropped` once more in the pre-split hp format."""
    T, D = x.shape
    dx = torch.empty_like(x)
    dropped = torch.empty_like(x) if dropout_p > 0 else None
    if pending is not None and len(pending) >= 16:
        layernorm_bwd_flush(pending)
    ws = _ln_workspace(D, x.device, pending)
    defer = pending is not None
    slabs, slab_stride = (dy.n, dy.stride) if isinstance(dy, Slabs) else (1, 0)    # (linear_bwd_input: defer_reduce)
    check(lib().ptamd_layernorm_bwd_dropout(ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), T, D, float(dropout_p),
                                            int(seed), int(stream_id), ptr(dx), ptr(dropped), ptr(row_scale),
                                            ptr(bound_factor), ptr(bound_scale), ptr(row_scale_min), ptr(bound_scale_min),
                                            ptr(planes), None if defer else ptr(dgamma), None if defer else ptr(dbeta),
                                            slabs, slab_stride, ptr(ws), ws.numel(),
                                            stream()), "layernorm_bwd_dropout")
    if defer:
        pending.append((ws, D, dgamma, dbeta))
    return dx, (dropped if dropped is not None else dx)


def weight_scales(jobs):
    """jobs: list of dict(w=tensor view [rows, cols] (row stride ld), row_scale=, col_scale=, stats=) -> ptamd_weight_scales."""
    from ._lib import WScaleJob
    for i in range(0, len(jobs), 40):
        chunk = jobs[i:i + 40]
        arr = (WScaleJob * len(chunk))()
        for k, j in enumerate(chunk):
            w = j["w"]
            rows, cols = (w.shape[0], w.shape[1]) if w.dim() == 2 else (1, w.shape[0])
            arr[k] = WScaleJob(w=w.data_ptr(), rows=rows, cols=cols, ld=w.stride(0) if w.dim() == 2 else cols,
                               row_scale=j["row_scale"].data_ptr() if j.get("row_scale") is not None else None,
                               col_scale=j["col_scale"].data_ptr() if j.get("col_scale") is not None else None,
                               stats=j["stats"].data_ptr() if j.get("stats") is not None else None,
                               rows_only=int(bool(j.get("rows_only", False))))
        check(lib().ptamd_weight_scales(arr, len(chunk), stream()), "weight_scales")


def bound_scales(jobs):
    """jobs: list of dict(ln_gamma=, ln_beta=, w=, w_index=, bias=, sqrt_d=, post_scale=, out_scale=, out_value=) of stats
    records / outputs (tensors) -> ptamd_bound_scales."""
    from ._lib import BoundJob
    P = lambda t: t.data_ptr() if t is not None else None                     # noqa: E731
    for i in range(0, len(jobs), 40):
        chunk = jobs[i:i + 40]
        arr = (BoundJob * len(chunk))()
        for k, j in enumerate(chunk):
            arr[k] = BoundJob(ln_gamma_stats=P(j.get("ln_gamma")), ln_beta_stats=P(j.get("ln_beta")), w_stats=P(j["w"]),
                              w_stat_index=int(j["w_index"]), bias_stats=P(j.get("bias")), sqrt_d=float(j.get("sqrt_d", 0.0)),
                              post_scale=float(j.get("post_scale", 1.0)), out_scale=P(j.get("out_scale")),
                              out_value=P(j.get("out_value")))
        check(lib().ptamd_bound_scales(arr, len(chunk), stream()), "bound_scales")


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=None, pending=None):
    """dx = LayerNorm'(dy) (+ dres, the gradient that bypassed the sublayer through the residual add).  `pending`: a list
    that collects the (dgamma, dbeta) reduction of this site for `layernorm_bwd_flush` instead of launching it here."""
    T, D = x.shape
    dx = torch.empty_like(x)
    if pending is not None and len(pending) >= 16:
        layernorm_bwd_flush(pending)
    ws = _ln_workspace(D, x.device, pending)
    defer = pending is not None
    check(lib().ptamd_layernorm_bwd(ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), T, D, ptr(dx),
                                    None if defer else ptr(dgamma), None if defer else ptr(dbeta), ptr(ws), ws.numel(), stream()),
          "layernorm_bwd")
    if defer:
        pending.append((ws, D, dgamma, dbeta))
    return dx


def embed_fwd(seq, emb, pe, dropout_p, seed):
    B, L = seq.shape
    D = emb.shape[1]
    out = torch.empty(B * L, D, dtype=torch.float32, device=emb.device)
    check(lib().ptamd_embed_fwd(ptr(seq), ptr(emb), ptr(pe), B, L, D, float(dropout_p), int(seed), ptr(out), stream()),
          "embed_fwd")
    return out


def embed_bwd(seq, dout, D, dropout_p, seed, demb):
    B, L = seq.shape
    ws = workspace("emb", lib().ptamd_embed_bwd_workspace_bytes(D), dout.device)
    check(lib().ptamd_embed_bwd(ptr(seq), ptr(dout), B, L, D, float(dropout_p), int(seed), ptr(demb), ptr(ws),
                                ws.numel(), stream()), "embed_bwd")


def attention_bwd_reads_keep_bits(B, L, H, dk, arith):
    return bool(lib().ptamd_attention_bwd_reads_keep_bits(B, L, H, dk, int(_DEFAULT_ARITH if arith is None else arith)))


def attention_keep_bits(B, L, H, device):
    """Buffer for the dropout decisions ptamd_attention_fwd hands to ptamd_attention_bwd (f16x2 arithmetic, dk 32 / 64)."""
    return torch.empty(lib().ptamd_attention_keep_bits_bytes(B, L, H) // 4, dtype=torch.int32, device=device)


def attention_reads_kv_planes(B, L, H, dk, arith):
    """True when the attention kernels of this shape and arithmetic read K / V pre-split (written by the QKV product's epilogue)."""
    return bool(lib().ptamd_attention_reads_kv_planes(B, L, H, dk, int(_DEFAULT_ARITH if arith is None else arith)))


def attention_kv_buffers(T, H, device):
    """(planes uint8, inverse group scales float32) for the pre-split K / V of T tokens and H heads of 64 (csrc/kv_format.h)."""
    return (torch.empty(lib().ptamd_attention_kv_bytes(T, H), dtype=torch.uint8, device=device),
            torch.empty(lib().ptamd_attention_kv_inv_floats(T, H), dtype=torch.float32, device=device))


def attention_fwd(qkv, seq, H, dropout_p, seed, stream_id, arith=None, keep_bits=None, kv=None):
    """keep_bits (attention_keep_bits, optional): filled with the dropout decisions when dropout_p > 0.  kv (optional):
    `attention_kv_buffers` filled by the QKV product - K and V are read from there, only the Q columns of `qkv` are."""
    B, L = seq.shape
    D = qkv.shape[1] // 3
    out = torch.empty(B * L, D, dtype=torch.float32, device=qkv.device)
    lse = torch.empty(B, H, L, dtype=torch.float32, device=qkv.device)
    check(lib().ptamd_attention_fwd(ptr(qkv), ptr(seq), B, L, H, D // H, float(dropout_p), int(seed), int(stream_id),
                                    int(_DEFAULT_ARITH if arith is None else arith), ptr(out), ptr(lse), ptr(keep_bits),
                                    ptr(kv[0]) if kv is not None else None, ptr(kv[1]) if kv is not None else None,
                                    stream()), "attention_fwd")
    return out, lse


def attention_bwd(qkv, seq, out, dout, lse, H, dropout_p, seed, stream_id, arith=None, row_scale=None, row_scale_min=None,
                  keep_bits=None, kv=None):
    """row_scale [T] / row_scale_min [4] (int32, preset to 0x7F000000): f16x2 scales of the rows of dqkv as a by-product
    (f16x2 arithmetic and head size 32 / 64 only - `attention_row_scales_available`).  keep_bits: what attention_fwd filled
    for the same (seed, stream_id) - the fused backward kernel reads the decisions instead of drawing them again."""
    B, L = seq.shape
    D = qkv.shape[1] // 3
    dqkv = torch.empty_like(qkv)
    ws = workspace("attn", lib().ptamd_attention_workspace_bytes(B, L, H, D // H), qkv.device)
    check(lib().ptamd_attention_bwd(ptr(qkv), ptr(seq), ptr(out), ptr(dout), ptr(lse), B, L, H, D // H,
                                    float(dropout_p), int(seed), int(stream_id),
                                    int(_DEFAULT_ARITH if arith is None else arith), ptr(dqkv), ptr(row_scale),
                                    ptr(row_scale_min), ptr(keep_bits), ptr(kv[0]) if kv is not None else None,
                                    ptr(kv[1]) if kv is not None else None, ptr(ws), ws.numel(), stream()), "attention_bwd")
    return dqkv


def attention_row_scales_available(dk, arith):
    return dk in (32, 64) and int(arith) in (GEMM_AUTO, GEMM_F16X2)


def relu_dropout_bwd(dy, y, dropout_p):
    dx = torch.empty_like(dy)
    check(lib().ptamd_relu_dropout_bwd(ptr(dy), ptr(y), dy.numel(), float(dropout_p), ptr(dx), stream()),
          "relu_dropout_bwd")
    return dx


def tanh_bwd(dy, y):
    dx = torch.empty_like(dy)
    check(lib().ptamd_tanh_bwd(ptr(dy), ptr(y), dy.numel(), ptr(dx), stream()), "tanh_bwd")
    return dx


def dropout_bwd(dy, dropout_p, seed, stream_id):
    rows, cols = dy.shape
    dx = torch.empty_like(dy)
    check(lib().ptamd_dropout_bwd(ptr(dy), rows, cols, float(dropout_p), int(seed), int(stream_id), ptr(dx), stream()),
          "dropout_bwd")
    return dx


def grad_sqnorm(g, out):
    ws = workspace("sqnorm", lib().ptamd_grad_sqnorm_workspace_bytes(), g.device)
    check(lib().ptamd_grad_sqnorm(ptr(g), g.numel(), ptr(out), ptr(ws), ws.numel(), stream()), "grad_sqnorm")
    return out


def sgd_step(w, g, sqnorm, max_norm, lr, weight_decay, zero_grad=False):
    """`zero_grad`: g is zeroed behind its read (the next step's `optimizer.zero_grad()` folded into this pass)."""
    check(lib().ptamd_sgd_step(ptr(w), ptr(g), w.numel(), ptr(sqnorm), float(max_norm or 0.0), float(lr),
                               float(weight_decay), int(bool(zero_grad)), stream()), "sgd_step")


def adam_step(w, g, m, v, sqnorm, max_norm, lr, beta1, beta2, eps, weight_decay, step, zero_grad=False):
    check(lib().ptamd_adam_step(ptr(w), ptr(g), ptr(m), ptr(v), w.numel(), ptr(sqnorm), float(max_norm or 0.0),
                                float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                                int(bool(zero_grad)), stream()), "adam_step")


def fill_u32(jobs):
    """jobs: list of (int32 / uint32 / float32 tensor, value) -> every element of each tensor = value (its bit pattern), ONE
    launch for up to 8 contiguous tensors (ptamd_fill_u32)."""
    from ._lib import FillJob
    for i in range(0, len(jobs), 8):
        chunk = jobs[i:i + 8]
        arr = (FillJob * len(chunk))()
        for k, (t, value) in enumerate(chunk):
            assert t.is_contiguous() and t.element_size() == 4
            arr[k] = FillJob(dst=t.data_ptr(), n=t.numel(), value=int(value) & 0xFFFFFFFF)
        check(lib().ptamd_fill_u32(arr, len(chunk), stream()), "fill_u32")


# ----------------------------------------------------------------------------- conv-enc front end
def pad4(c):
    return (c + 3) // 4 * 4


def conv1d_fwd(x, B, L, C, w, bias, k, arith=None):
    """x [T, pad4(C)] token-major, w [Co, C, k] (torch Conv1d layout) -> y [T, Co]; returns (y, packed weight)."""
    T, Co, Cp = x.shape[0], w.shape[0], pad4(C)
    col = torch.empty(T, k * Cp, dtype=torch.float32, device=x.device)
    check(lib().ptamd_im2col1d(ptr(x), x.stride(0), B, L, Cp, k, ptr(col), stream()), "im2col1d")
    w2 = torch.empty(Co, k * Cp, dtype=torch.float32, device=x.device)
    check(lib().ptamd_conv_weight_pack(ptr(w), Co, C, k, ptr(w2), stream()), "conv_weight_pack")
    y = linear_fwd(col, w2, bias, arith=arith)
    return y, w2


def conv1d_bwd(dy, x, B, L, C, w2, k, dw, dbias, need_dx=True, arith=None):
    """Accumulates dw [Co, C, k] and dbias [Co]; returns dx [T, pad4(C)] (or None)."""
    T, Co, Cp = x.shape[0], dy.shape[1], pad4(C)
    col = torch.empty(T, k * Cp, dtype=torch.float32, device=x.device)          # recomputed, not stored
    check(lib().ptamd_im2col1d(ptr(x), x.stride(0), B, L, Cp, k, ptr(col), stream()), "im2col1d")
    dw2 = torch.zeros(Co, k * Cp, dtype=torch.float32, device=x.device)
    linear_bwd_weight(dy, col, dw2, arith=arith)
    check(lib().ptamd_conv_weight_unpack_add(ptr(dw2), Co, C, k, ptr(dw), stream()), "conv_weight_unpack_add")
    colsum(dy, dbias)
    if not need_dx:
        return None
    dcol = linear_bwd_input(dy, w2, arith=arith)
    dx = torch.empty(T, Cp, dtype=torch.float32, device=x.device)
    check(lib().ptamd_col2im1d(ptr(dcol), B, L, Cp, k, ptr(dx), dx.stride(0), stream()), "col2im1d")
    return dx


def onehot(seq, C):
    T = seq.numel()
    x = torch.empty(T, pad4(C), dtype=torch.float32, device=seq.device)
    check(lib().ptamd_onehot(ptr(seq), T, C, ptr(x), stream()), "onehot")
    return x


def posenc_add_fwd(x, pe, B, L, dropout_p, seed):
    y = torch.empty_like(x)
    check(lib().ptamd_posenc_add_fwd(ptr(x), ptr(pe), B, L, x.shape[1], float(dropout_p), int(seed), ptr(y), stream()),
          "posenc_add_fwd")
    return y


def posenc_add_bwd(dy, dropout_p, seed):
    dx = torch.empty_like(dy)
    check(lib().ptamd_posenc_add_bwd(ptr(dy), dy.numel(), float(dropout_p), int(seed), ptr(dx), stream()), "posenc_add_bwd")
    return dx


# ----------------------------------------------------------------------------- the weights of a step in one pass (csrc/wprep.hip)
class WeightsPrep:
    """Plan of ptamd_weights_prep / ptamd_sgd_step_prep / ptamd_adam_step_prep for one model: the device-resident tables of
    include/ptamd.h (`ptamd_wprep_plan`) built from a description of the flat parameter buffer.

      numel    floats in the flat buffer
      segs     list of dict(offset, rows, cols, stats_row0=0, stats=None | record index, row_scale=None | view of `ints`,
               col_scale=None | view of `ints`, colnorm=False, row_planes=None | uint8 tensor, col_planes=None | uint8 tensor)
               - pairwise disjoint pieces of the buffer; everything else is updated as plain parameters
      groups   list (an encoder layer each) of lists of dict(ln_gamma, ln_beta, w, w_index, bias [statistics records or
               None], sqrt_d, post_scale, out_scale [view of `ints`] / out_value [view of `values`])
      ints     int32 tensor that holds every scale; values: float tensor that holds every bound value; nstats: records
    """

    def __init__(self, numel, segs, groups, ints, values, nstats):
        import numpy as np
        from ._lib import WprepBound, WprepPlan, WprepSeg
        dev = ints.device
        RB, PF = lib().ptamd_wprep_rows_per_block(), lib().ptamd_wprep_plain_floats_per_block()
        PANEL = 512                     # kernel A keeps whole rows in registers: wider matrices go through it as column panels
        segs = sorted(segs, key=lambda s: s["offset"])
        idx = lambda view, base: -1 if view is None else view.storage_offset() - base.storage_offset()       # noqa: E731
        table, blocks_a, blocks_b, plain = [], [], [], []
        ncolmax = ncolsq = 0            # entries of the pool of maxima (column maxima, row maxima of panelled matrices) / of colsq
        pos = 0
        for s in segs:
            rows, cols, off = int(s["rows"]), int(s["cols"]), int(s["offset"])
            if cols % 4 or off % 4 or off < pos or (cols > PANEL and cols % PANEL):
                raise ValueError(f"ptamd_weights_prep: segment (offset {off}, {rows} x {cols}) is not representable")
            if off > pos:
                plain.append((pos, off - pos))
            pos = off + rows * cols
            want_cols = s.get("col_scale") is not None
            for name in ("row_planes", "col_planes"):
                if s.get(name) is not None and (rows % 32 or cols % 32):
                    raise ValueError("ptamd_weights_prep: planes need rows and cols that are multiples of 32")
            if s.get("col_planes") is not None and not want_cols:
                raise ValueError("ptamd_weights_prep: col_planes need col_scale")
            if s.get("colnorm") and (s.get("stats") is None or not want_cols):
                raise ValueError("ptamd_weights_prep: colnorm needs a statistics record and col_scale")
            stats = -1 if s.get("stats") is None else int(s["stats"])
            npanels = -(-cols // PANEL)
            wide = npanels > 1
            if wide and (s.get("row_planes") is not None or int(s.get("stats_row0", 0)) != 0):
                raise ValueError("ptamd_weights_prep: a matrix wider than 512 columns cannot have row planes or a statistics sub-range")
            colmax0 = ncolmax
            ncolmax += cols if want_cols else 0
            rowmax0 = -1
            if wide and s.get("row_scale") is not None:
                rowmax0 = ncolmax
                ncolmax += rows
            nrb = -(-rows // RB)
            for pnl in range(npanels):
                pc = min(PANEL, cols - pnl * PANEL)
                k = len(table)
                table.append(WprepSeg(
                    offset=off + pnl * PANEL, rows=rows, cols=pc, stats_row0=-1 if wide else int(s.get("stats_row0", 0)),
                    stats_index=stats, row_scale_index=-1 if wide else idx(s.get("row_scale"), ints),
                    col_scale_index=idx(s.get("col_scale"), ints) + pnl * PANEL if want_cols else -1,
                    colmax_index=colmax0 + pnl * PANEL if want_cols else 0,
                    colsq_index=ncolsq if s.get("colnorm") else -1,
                    row_planes=0 if s.get("row_planes") is None else s["row_planes"].data_ptr(), col_planes=0,
                    ld=cols, rowmax_index=rowmax0))
                blocks_a += [(k, b) for b in range(nrb)]
                if want_cols:
                    blocks_b.append((0, k, 0, 0))
                if s.get("colnorm"):
                    ncolsq += nrb * pc
            if wide and rowmax0 >= 0 or s.get("col_planes") is not None:
                # the matrix as a whole, for kernel B only (not in blocks_a): row scales of a panelled matrix, planes of its transpose
                k = len(table)
                table.append(WprepSeg(offset=off, rows=rows, cols=cols, stats_row0=0, stats_index=-1,
                                      row_scale_index=idx(s.get("row_scale"), ints) if wide else -1, col_scale_index=-1,
                                      colmax_index=colmax0, colsq_index=-1, row_planes=0,
                                      col_planes=0 if s.get("col_planes") is None else s["col_planes"].data_ptr(), ld=cols,
                                      rowmax_index=rowmax0))
                if rowmax0 >= 0:
                    blocks_b += [(3, k, b, 0) for b in range(-(-rows // 256))]
                if s.get("col_planes") is not None:
                    nchunks = (cols // 32) * (-(-rows // 32) * 2) * 64           # rows of W^T = cols; 16-column blocks of its K = rows
                    blocks_b += [(1, k, b, 0) for b in range(-(-nchunks // 256))]
        if pos < numel:
            plain.append((pos, numel - pos))
        arr = (WprepSeg * len(table))(*table)
        n_matrix_blocks = len(blocks_a)
        for j, (first, n) in enumerate(plain):
            blocks_a += [(-(j + 1), b) for b in range(-(-n // PF))]
        # bound jobs, per group; the segments whose column norm a group needs are those flagged `colnorm` whose statistics
        # record one of the group's jobs reads with w_index == 1
        by_stats = {}
        for k, t in enumerate(table):
            if t.colsq_index >= 0:
                by_stats.setdefault(int(t.stats_index), []).append(k)
        bounds, grp_rows, cn_list = [], [], []
        rec = lambda v: -1 if v is None else int(v)                                                         # noqa: E731
        for g, jobs in enumerate(groups):
            first, cn_first = len(bounds), len(cn_list)
            for j in jobs:
                bounds.append(WprepBound(ln_gamma_stats=rec(j.get("ln_gamma")), ln_beta_stats=rec(j.get("ln_beta")), w_stats=int(j["w"]),
                                         w_stat_index=int(j["w_index"]), bias_stats=rec(j.get("bias")),
                                         sqrt_d=float(j.get("sqrt_d", 0.0)), post_scale=float(j.get("post_scale", 1.0)),
                                         out_scale=idx(j.get("out_scale"), ints), out_value=idx(j.get("out_value"), values)))
                if int(j["w_index"]) == 1 and int(j["w"]) in by_stats:
                    cn_list += [k for k in by_stats[int(j["w"])] if k not in cn_list[cn_first:]]
            grp_rows.append((first, len(bounds) - first, cn_first, len(cn_list) - cn_first))
            blocks_b.append((2, g, 0, 0))
        if not groups:
            raise ValueError("ptamd_weights_prep: at least one bounds group (it also resets the statistics)")
        up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=dt))).to(dev)           # noqa: E731
        self.t_segs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.t_blocks_a = up(blocks_a, np.int32).reshape(-1, 2)
        self.t_plain = up(plain if plain else [(0, 0)], np.int64).reshape(-1, 2)
        self.t_blocks_b = up(blocks_b, np.int32).reshape(-1, 4)
        barr = (WprepBound * max(len(bounds), 1))(*bounds)
        self.t_bounds = torch.frombuffer(bytearray(bytes(barr)), dtype=torch.uint8).to(dev)
        self.t_groups = up(grp_rows, np.int32).reshape(-1, 4)
        self.t_colnorm = up(cn_list if cn_list else [0], np.int32)
        self.colmax = torch.zeros(2, max(ncolmax, 1), dtype=torch.int32, device=dev)
        self.colsq = torch.zeros(max(ncolsq, 2), dtype=torch.float64, device=dev)
        self.stats = torch.zeros(2, nstats, 4, dtype=torch.float32, device=dev)
        self.ints, self.values = ints, values
        self.parity = 0
        self.numel = int(numel)
        self.plan = WprepPlan(segs=self.t_segs.data_ptr(), nsegs=len(table), blocks_a=self.t_blocks_a.data_ptr(),
                              nblocks_a=len(blocks_a), nblocks_a_matrices=n_matrix_blocks, plain=self.t_plain.data_ptr(),
                              nplain=len(plain), blocks_b=self.t_blocks_b.data_ptr(), nblocks_b=len(blocks_b),
                              bounds=self.t_bounds.data_ptr(), nbounds=len(bounds), groups=self.t_groups.data_ptr(),
                              ngroups=len(grp_rows), colnorm_segs=self.t_colnorm.data_ptr(), scales=ints.data_ptr(),
                              values=values.data_ptr(), colmax=self.colmax.data_ptr(), ncolmax=max(ncolmax, 1),
                              colsq=self.colsq.data_ptr(), stats=self.stats.data_ptr(), nstats=int(nstats), numel=int(numel),
                              with_planes=1)

    def _call(self, what, fn, with_planes, *args):
        """One library call on the copy of the double-buffered maxima / statistics that is due; the parity moves only once the
        call has been accepted (a refused call - bad argument, launch error - must not leave the next one accumulating into
        the copy that was never reset)."""
        self.plan.with_planes = int(bool(with_planes))
        check(fn(C.byref(self.plan), *[self.parity if a is _PARITY else a for a in args]), what)
        self.parity ^= 1

    def last_stats(self):
        """The statistics records of the most recent call ([nstats, 4] view)."""
        return self.stats[self.parity ^ 1]

    def prepare(self, flat, with_planes=True):
        """Scales, statistics, bounds (and planes) of the weights as they are: no update."""
        assert flat.numel() == self.numel
        self._call("weights_prep", lib().ptamd_weights_prep, with_planes, ptr(flat), _PARITY, stream())

    def sgd_step(self, w, g, sqnorm, max_norm, lr, weight_decay, with_planes=True, zero_grad=False):
        self._call("sgd_step_prep", lib().ptamd_sgd_step_prep, with_planes, _PARITY, ptr(w), ptr(g), w.numel(), ptr(sqnorm),
                   float(max_norm or 0.0), float(lr), float(weight_decay), int(bool(zero_grad)), stream())

    def adam_step(self, w, g, m, v, sqnorm, max_norm, lr, beta1, beta2, eps, weight_decay, step, with_planes=True,
                  zero_grad=False):
        self._call("adam_step_prep", lib().ptamd_adam_step_prep, with_planes, _PARITY, ptr(w), ptr(g), ptr(m), ptr(v), w.numel(),
                   ptr(sqnorm), float(max_norm or 0.0), float(lr), float(beta1), float(beta2), float(eps),
                   float(weight_decay), int(step), int(bool(zero_grad)), stream())


_PARITY = object()       # placeholder of WeightsPrep._call: "the copy that is due"
