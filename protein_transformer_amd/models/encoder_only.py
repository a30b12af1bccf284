"""`-m enc-only`: the encoder-only Transformer of the reference, executed by libptamd kernels.

Drop-in for /root/reference/protein_transformer/models/encoder_only.py:10-45 (constructor signature,
`.forward(enc_input, dec_input=None)`, `.predict`, `_init_parameters`) with the reference's module tree
(models/transformer/Encoder.py:8-54, Attention.py:24-69, Sublayers.py:5-72) kept only as a *naming*
skeleton, so `state_dict()` has exactly the reference's keys and checkpoints load
either way.  No torch op computes anything here:

  * all parameters are views into ONE flat fp32 buffer (wq|wk|wv adjacent = one fused [3D,D] GEMM operand;
    the flat buffer is what the fused optimizer and the RCCL gradient all-reduce work on);
  * forward and backward are one autograd.Function that enqueues the HIP kernels (embedding, LayerNorm,
    fp32-MFMA GEMMs with fused bias/ReLU/dropout/residual/tanh epilogues, fused masked attention) and
    writes gradients straight into one flat gradient buffer;
  * dropout masks are counter-based (seed, site) and regenerated in the backward pass.

Reference quirks reproduced on purpose: the embedding is added twice
(Encoder.py:30 + Sublayers.py:60), attention-probability dropout is 0.1 regardless of `dropout`
(Attention.py:31), there is no final LayerNorm, pad id 20 has an ordinary embedding row, and the
output layer starts as weight 0 / bias arctanh(angle_means) (encoder_only.py:28-34).
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from .. import kernels as K
from ..protein.Structure import NUM_PREDICTED_ANGLES

# dropout sites inside one encoder layer (stream ids = layer * 8 + site)
_SITE_ATTN, _SITE_ATTN_OUT, _SITE_FFN_HID, _SITE_FFN_OUT = 0, 1, 2, 3


class _Holder(nn.Module):
    """Names parameters like the reference's Linear / LayerNorm / Embedding modules; computes nothing."""

    def __init__(self, **shapes):
        super().__init__()
        for name, shape in shapes.items():
            self.register_parameter(name, nn.Parameter(torch.empty(*shape)))


class _AttnHolder(nn.Module):
    def __init__(self, dm):
        super().__init__()
        for nm in ("wq", "wk", "wv", "wo"):
            setattr(self, nm, _Holder(weight=(dm, dm), bias=(dm,)))


class _PwffHolder(nn.Module):
    def __init__(self, dm, dff):
        super().__init__()
        self.layer1 = _Holder(weight=(dff, dm), bias=(dff,))
        self.layer2 = _Holder(weight=(dm, dff), bias=(dm,))


class _SublayerHolder(nn.Module):
    def __init__(self, dm):
        super().__init__()
        self.norm = _Holder(weight=(dm,), bias=(dm,))


class _LayerHolder(nn.Module):
    def __init__(self, dm, dff):
        super().__init__()
        self.self_attn = _AttnHolder(dm)
        self.pwff = _PwffHolder(dm, dff)
        self.sublayer_connections = nn.ModuleList([_SublayerHolder(dm) for _ in range(2)])


class _EmbHolder(nn.Module):
    def __init__(self, vocab, dm):
        super().__init__()
        self.emb = _Holder(weight=(vocab, dm))


class _PeHolder(nn.Module):
    def __init__(self, dm, max_seq_len):
        super().__init__()
        # Sublayers.py:48-56
        pe = torch.zeros(max_seq_len, dm)
        position = torch.arange(0., max_seq_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0., dm, 2) * -(np.log(10000.0) / dm))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe.unsqueeze(0))


class _EncoderHolder(nn.Module):
    def __init__(self, din, dm, dlayer, dff, n_layers, max_seq_len, conv_shapes, use_embedding):
        super().__init__()
        if use_embedding:
            self.input_embedding = _EmbHolder(din, dm)
            self.positional_enc = _PeHolder(dm, max_seq_len)
        else:
            self.positional_enc = _PeHolder(dlayer, max_seq_len)
        if conv_shapes is not None:      # the reference's conv-enc has this (possibly empty) list, enc-only does not
            self.conv_layers = nn.ModuleList([_Holder(weight=(co, ci, k), bias=(co,)) for ci, co, k in conv_shapes])
        self.enc_layers = nn.ModuleList([_LayerHolder(dlayer, dff) for _ in range(n_layers)])


class AutoGuard:
    """Run-time guard of the AUTO arithmetic's two assumptions (csrc/scales.hip).

    (1) BOUND-derived scales.  Five operands per encoder layer are scaled by a bound that follows from the weights alone
    instead of their measured maximum: the attention output `att` and the FFN hidden layer `f1` (A of the two products
    behind them and one operand of their weight-gradient products), the hidden gradient `dz1` (bound from the row norms of
    `dy2`), and the LayerNorm outputs `h1`, `h2` as operands of weight-gradient products.  A bound 2^k above the true
    maximum costs k of the 18 binades in which an element keeps its 22 bits: harmless while k is small, garbage at k = 18
    (measured: LayerNorm gains spanning 2^+-8 against compensating weight columns give k = 17 .. 19 and predictions that are
    0.3 off, tests/test_gpu_auto_guard.py).  Nothing inside a step measures k - the guard does, off the hot path.
    (2) Moderate dynamic range ALONG THE CONTRACTED INDEX.  A two-term f16 product is accurate relative to the largest
    element of an operand row; a function-preserving rescaling of hidden units (row n of W1 times 2^c, column n of W2 times
    2^-c) spreads both operands of the product over c binades along n.  The per-row / per-column scales of every weight
    matrix are computed every step anyway: their spread (binades between the largest and the smallest) along the
    contracted index of each activation x weight product is the second thing the guard looks at.

    Mechanics: every `interval`-th training step (and the first one) the backward pass measures the true max |x| of the five
    operands of every layer (ptamd_weight_scales' statistics: one streaming launch per layer, ~0.35 ms per measured step at
    config 4) and copies them, the bound scales and the weight scales to pinned host memory behind an event; the next
    forward pass that finds the event complete turns them into `slack[layer][site]` and `spread[layer][product]` - no host
    synchronisation anywhere.  A site whose slack exceeds `max_slack` (or whose bound was VIOLATED, slack < 0) stops using
    its bound: the activation x weight products take the exact row scales of a pass over the operand (ptamd_gemm finds
    them itself when `a_scale` is NULL - still f16x2, exact scales), the weight-gradient products run in bf16x3 (exact
    three-term split, no scales).  A product whose weight-scale spread exceeds `max_spread` runs in bf16x3.  UNTIL THE FIRST
    MEASUREMENT HAS LANDED NOTHING IS TRUSTED: every site is off its bound and every product counts as wide (the first step
    of a model, or the first steps while the copy is in flight, cost what the bf16x3 arithmetic costs).
    `fallback_products` counts the products switched, per step.

    Determinism (round 5): a measurement is honoured at a FIXED forward pass - the second one after the backward pass that
    submitted it (the next one for a measurement taken in a forward pass) - behind `event.synchronize()`, not whenever the
    asynchronous copy happens to have landed.  By then the event is long complete (the host has waited for the loss
    report of the step in between), so the wait costs nothing, and the step at which the arithmetic changes no longer
    depends on host timing: identical seeds give bit-identical trajectories and data-parallel ranks switch together.
    `reset()` (load_state_dict, a rebuilt flat buffer, in-place weight surgery) forgets everything that was measured on
    the old weights: nothing is trusted again until a fresh measurement has been honoured."""
    SITES = ("att", "f1", "dz1", "h1", "h2")
    # products a site switches when it falls back: (activation x weight products, weight-gradient products)
    PRODUCTS = {"att": 2, "f1": 2, "dz1": 2, "h1": 1, "h2": 1}
    # activation x weight products and the weight scales along their contracted index (names of _step_scales)
    WIDE = (("qkv", "cs_qkv"), ("wo", "cs_o"), ("ff1", "cs_1"), ("ff2", "cs_2"),
            ("dx_qkv", "rs_qkv"), ("dx_wo", "rs_o"), ("dx_ff1", "rs_1"), ("dx_ff2", "rs_2"))

    def __init__(self, nlayers, interval=16, max_slack=8, max_spread=12):
        self.nlayers, self.interval, self.max_slack, self.max_spread = nlayers, int(interval), int(max_slack), int(max_spread)
        self.enabled = True
        self.off = np.ones((nlayers, len(self.SITES)), dtype=bool)      # True: the site does not use its bound
        self.wide = np.ones((nlayers, len(self.WIDE)), dtype=bool)      # True: the product runs in bf16x3
        self.slack = self.spread = None                                 # last measurement [nlayers, 5] / [nlayers, 8]
        self.max_slack_seen = np.full(len(self.SITES), -np.inf)
        self.max_spread_seen = -np.inf
        self.violations = 0                                             # bounds found BELOW the measured maximum
        self.train_steps = self.measured_steps = self.fallback_products = 0
        self.passes = self.eval_passes = 0                              # forward passes polled / of them without a backward pass
        self._pending = None
        self._force = False

    def reset(self):
        """The weights were replaced (load_state_dict, a new flat buffer, edits by hand): what was measured on the old ones
        says nothing about these.  Back to the untrusting state; the next pass measures."""
        self.off[:] = True
        self.wide[:] = True
        self.slack = self.spread = None
        self.measured_steps = 0
        self._pending = None
        self._force = True

    def want_measure(self):
        return self.enabled and self._pending is None and (self._force or self.train_steps % max(self.interval, 1) == 0)

    def want_measure_forward(self):
        """A pass that no backward pass follows (evaluation, inference): measure the four forward operands while nothing has
        been measured on these weights, and again every `interval`-th such pass (the inputs change even if the weights do
        not: the slack of a bound is a property of both)."""
        return self.enabled and self._pending is None and (self._force or self.measured_steps == 0
                                                           or self.eval_passes % max(self.interval, 1) == 0)

    def submit(self, stats, ints, minbuf, layers, forward_only=False):
        """Called at the end of a measuring backward pass (or of a measuring evaluation pass, `forward_only`): asynchronous
        copies of the measured maxima and of the bound / weight scales into pinned memory, one event behind them; honoured
        by `poll` at a fixed later pass."""
        # (the pinned buffers are kept: a measurement is only submitted once the previous one has been read, and a pinned
        # allocation costs a few hundred microseconds of host time inside a step)
        host = self.__dict__.get("_host")
        if host is None or any(h.shape != t.shape or h.dtype != t.dtype for h, t in zip(host, (stats, ints, minbuf))):
            host = self._host = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in (stats, ints, minbuf)]
        for h, t in zip(host, (stats, ints, minbuf)):
            h.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        # element ranges of the scales inside `ints` (views of it, see _step_scales)
        names = ("att_scale", "f1_scale", "h1_scale", "h2_scale") + tuple(w for _, w in self.WIDE)
        where = [{k: (L[k].storage_offset() - ints.storage_offset(), L[k].numel()) for k in names} for L in layers]
        self._force = False
        self._pending = (ev, host, where, self.passes + (1 if forward_only else 2), bool(forward_only))

    def settle(self):
        """Honour a submitted measurement NOW (waits for its event): for callers that want the steady state at a point of
        their choosing - tests and parity records that compare "the pass after the first measurement" - instead of at the
        pass it is due.  Deterministic like `poll`: what is applied does not depend on when it is applied.  Returns True
        when there was one."""
        if self._pending is None:
            return False
        self._apply()
        return True

    def poll(self):
        """Called at the start of every forward pass: a submitted measurement is honoured at the pass it is DUE (see the
        class comment), behind a wait for its event - never earlier, however soon the copy landed."""
        self.passes += 1
        if self._pending is None or self.passes < self._pending[3]:
            return
        self._apply()

    def _apply(self):
        ev, (stats, ints, minbuf), where, _, forward_only = self._pending
        ev.synchronize()
        self._pending = None
        mx = stats.numpy()[:, :, 2].astype(np.float64)                  # [nlayers, 5]: max |x| of att, f1, dz1, h1, h2
        bits = ints.numpy().view(np.uint32)
        mb = minbuf.numpy().view(np.uint32)
        sb = np.array([[bits[where[i]["att_scale"][0]], bits[where[i]["f1_scale"][0]], mb[i, 4], bits[where[i]["h1_scale"][0]],
                        bits[where[i]["h2_scale"][0]]] for i in range(self.nlayers)], dtype=np.uint32)
        slack = self.slack_binades(mx, sb)
        if forward_only and self.slack is not None:
            slack[:, 2] = self.slack[:, 2]          # dz1 exists in backward passes only: keep what the last one measured
        spread = np.array([[self.spread_binades(bits[where[i][w][0]:where[i][w][0] + where[i][w][1]]) for _, w in self.WIDE]
                           for i in range(self.nlayers)])
        self.slack, self.spread = slack, spread
        self.measured_steps += 1
        self.max_slack_seen = np.maximum(self.max_slack_seen, slack.max(0))
        self.max_spread_seen = max(self.max_spread_seen, float(spread.max()))
        self.violations += int((slack < 0).sum())
        self.off = (slack > self.max_slack) | (slack < 0)
        self.wide = spread > self.max_spread

    @staticmethod
    def spread_binades(scale_bits):
        """Binades between the largest and the smallest of a weight matrix's row (column) maxima, from their f16x2 scales
        (powers of two; all-zero rows - the largest finite scale - do not count)."""
        e = (np.asarray(scale_bits, dtype=np.uint32) >> 23).astype(np.int64)
        e = e[e < 254]
        return float(e.max() - e.min()) if e.size else 0.0

    @staticmethod
    def slack_binades(max_abs, scale_bits):
        """Binades by which a bound exceeds what it bounds.  `scale_bits`: the f16x2 scale derived from the bound - the
        power of two that takes the BOUND into [2^14, 2^15) - as uint32 bit patterns; `max_abs`: the measured max |x|.
        max_abs * scale lies in [2^(14-k), 2^(15-k)): k = 0 when the bound is as good as the maximum, k < 0 when the bound
        was too small (the f16 split would overflow from k <= -1).  Slots whose scale is the atomicMin preset 0x7F000000
        (no bound-derived scale in this pass) and all-zero operands count as k = 0."""
        max_abs = np.asarray(max_abs, dtype=np.float64)
        scale_bits = np.asarray(scale_bits, dtype=np.uint32)
        s = scale_bits.view(np.float32).astype(np.float64)
        m = max_abs * s
        ok = (scale_bits != 0x7F000000) & np.isfinite(m) & (m > 0)
        with np.errstate(divide="ignore", invalid="ignore"):
            k = 14.0 - np.floor(np.log2(np.where(ok, m, 2.0 ** 14)))
        return np.where(ok, k, 0.0)

    def count_step(self):
        self.train_steps += 1
        self.fallback_products += int(sum(self.PRODUCTS[s] * int(self.off[:, j].sum()) for j, s in enumerate(self.SITES)))
        self.fallback_products += int(self.wide.sum())

    def report(self):
        return {"fallbacks_per_step": self.fallback_products / max(self.train_steps, 1), "measured_steps": self.measured_steps,
                "steps": self.train_steps, "interval": self.interval, "max_slack_binades_allowed": self.max_slack,
                "max_weight_scale_spread_binades_allowed": self.max_spread, "bound_violations": self.violations,
                "max_slack_binades_seen": {s: (None if not np.isfinite(v) else float(v))
                                           for s, v in zip(self.SITES, self.max_slack_seen)},
                "max_weight_scale_spread_binades_seen": None if not np.isfinite(self.max_spread_seen) else self.max_spread_seen,
                "sites_off_bounds_now": int(self.off.sum()), "products_in_bf16x3_now": int(self.wide.sum())}


class _TransformerBase(nn.Module):
    """Shared implementation of `enc-only` and `conv-enc`: parameter flattening, forward/backward driver."""

    def __init__(self, nlayers, nhead, dmodel, dff, max_seq_len, vocab, angle_means, use_tanh_out, dropout=0.1,
                 conv_kernel_sizes=None, conv_dim_reductions=(), use_embedding=True, conv_out_matches_dm=True):
        super().__init__()
        self.angle_means = angle_means
        self.vocab = vocab
        self.nlayers, self.nhead, self.dmodel, self.dff, self.max_seq_len = nlayers, nhead, dmodel, dff, max_seq_len
        self.use_tanh_out = use_tanh_out
        self.use_embedding = bool(use_embedding)
        self.dropout = float(dropout)
        self.attn_dropout = 0.1                      # hard-wired in the reference (Attention.py:31, Encoder.py:47)
        # conv stack geometry (convolutional_encoder.py:85-104)
        self.conv_shapes = None
        din = dmodel if self.use_embedding else len(vocab)
        self.dlayer = dmodel
        if conv_kernel_sizes is not None:
            self.conv_shapes = []
            ks, drs = list(conv_kernel_sizes), list(conv_dim_reductions)
            for i, (k, dr) in enumerate(zip(ks, drs)):
                assert k % 2 != 0, "Kernel size must be odd to maintain sequence length."
                dout = dmodel if (i == len(ks) - 1 and conv_out_matches_dm) else int(din // dr)
                self.conv_shapes.append((din, dout, k))
                din = dout
            if conv_out_matches_dm:
                self.dlayer = dmodel
            else:
                d = dmodel if self.use_embedding else len(vocab)
                for dr in drs:
                    d /= dr
                self.dlayer = int(d)
        D = self.dlayer
        assert D % nhead == 0, "The dimension of the model must be evenly divisible by the number of attn heads."
        widths = [dmodel, D, dff] + [c for _, c, _ in (self.conv_shapes or [])]
        if any(w % 4 for w in widths) or (D // nhead) not in (8, 16, 32, 64):
            raise ValueError("libptamd needs channel widths that are multiples of 4 and width / n_head in {8, 16, 32, 64}")
        self.encoder = _EncoderHolder(len(vocab), dmodel, D, dff, nlayers, max_seq_len, self.conv_shapes,
                                      self.use_embedding)
        self.output_projection = _Holder(weight=(NUM_PREDICTED_ANGLES * 2, D), bias=(NUM_PREDICTED_ANGLES * 2,))
        self._flat = None
        self._flat_grad = None
        self._layout = self._make_layout()
        self.dropout_seed = 0x5DEECE66D
        self._step_counter = 0
        self.grad_hook = None                        # called with (offset, numel) as soon as a gradient slice is final
        self.attn_row_scales = True                  # dqkv row scales from the attention backward kernels (False: a pass; ablation)
        self.attn_mode = None                        # arithmetic of the attention kernels alone (ablations); None = gemm_mode
        self.gemm_mode = None                        # kernels.GEMM_* arithmetic of THIS model; None = kernels.get_gemm_mode()
        self.hp_forward = True                       # FFN-layer-1 forward product on ptamd_gemm_hp from LayerNorm-written planes (read every pass)
        self.hp_qkv = True                           # ... the QKV product too (three-stage kernel of round 4)
        self.keep_attn_bits = True                   # attention dropout decisions handed from the forward to the fused backward kernel
        self.ffn_gate_mask = True                    # FFN layer 1 leaves `f1 > 0` as 1 bit / element for the gated dX product of layer 2
        self.hp_dx = True                            # dX of FFN layer 2 there too (A: planes from the fused LayerNorm backward, B: W2^T planes): +-0 in
                                                     # the step while its gate was the fp32 activation, -0.06 ms with the 1-bit gate (NOTES section 12)
        self.side_stream_dw = True                   # small batches: weight-gradient products on a side stream
        self.top_layer_scales = True                 # uniform scales of the top layer's dy2 / dz1 by a pass (backward())
        self.dw_group = "auto"                       # grouping of the weight-gradient products of a layer (backward())
        self.kv_planes = True                        # the QKV product's epilogue writes K / V pre-split for the attention kernels (csrc/kv_format.h)
        self.weights_prep = True                     # scales / bounds / planes of the weights in one pass (csrc/wprep.hip), fused into the
                                                     # optimizer step where one precedes the forward pass; False: the separate launches of rounds 2-4
        self.auto_guard = AutoGuard(nlayers)         # measures the slack of the bound-derived f16x2 scales, falls back per site
        self._init_parameters()

    # ------------------------------------------------------------------ parameters
    def _make_layout(self):
        """name -> (offset, shape) in the flat buffer; every offset is a multiple of 4 floats (16 B)."""
        order = ["encoder.input_embedding.emb.weight"] if self.use_embedding else []
        for j in range(len(self.conv_shapes or [])):
            order += [f"encoder.conv_layers.{j}.weight", f"encoder.conv_layers.{j}.bias"]
        for i in range(self.nlayers):
            b = f"encoder.enc_layers.{i}."
            order += [b + f"self_attn.{n}.weight" for n in ("wq", "wk", "wv")]
            order += [b + f"self_attn.{n}.bias" for n in ("wq", "wk", "wv")]
            order += [b + "self_attn.wo.weight", b + "self_attn.wo.bias",
                      b + "pwff.layer1.weight", b + "pwff.layer1.bias", b + "pwff.layer2.weight", b + "pwff.layer2.bias",
                      b + "sublayer_connections.0.norm.weight", b + "sublayer_connections.0.norm.bias",
                      b + "sublayer_connections.1.norm.weight", b + "sublayer_connections.1.norm.bias"]
        order += ["output_projection.weight", "output_projection.bias"]
        params = dict(self.named_parameters())
        assert set(order) == set(params)
        layout, off = {}, 0
        for name in order:
            shape = tuple(params[name].shape)
            layout[name] = (off, shape)
            off += (int(np.prod(shape)) + 3) // 4 * 4
        self._flat_numel = off
        return layout

    def _init_parameters(self):
        # encoder_only.py:24-34: xavier_uniform on every >=2-D parameter, module defaults elsewhere
        # (nn.Linear bias ~ U(+-1/sqrt(fan_in)), LayerNorm 1 / 0), then the output layer
        for name, p in self.named_parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
            elif "norm.weight" in name:
                nn.init.ones_(p)
            elif "norm.bias" in name:
                nn.init.zeros_(p)
            elif "conv_layers" in name:
                ci, _, k = self.conv_shapes[int(name.split(".")[2])]
                nn.init.uniform_(p, -1 / math.sqrt(ci * k), 1 / math.sqrt(ci * k))
            else:
                fan_in = self.dff if "layer2" in name else self.dlayer
                nn.init.uniform_(p, -1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
        # encoder_only.py:28-33 / convolutional_encoder.py:33-36: `-m conv-enc-linear-out` (use_tanh_out=False, reachable
        # through train.py:289-298) starts the bias at the angle means themselves, there is no tanh to undo
        am = np.asarray(self.angle_means, dtype=np.float64)
        if self.use_tanh_out:
            am = np.arctanh(am)
        with torch.no_grad():
            self.output_projection.bias.copy_(torch.tensor(am, dtype=torch.float32))
            self.output_projection.weight.zero_()

    def _ensure_flat(self):
        """(Re)build the flat parameter / gradient buffers on the parameters' device and alias every
        named parameter (and its .grad) into them.  Idempotent; survives .to(device) and load_state_dict.
        Called several times per step: the fast path checks the two ends of the layout only (a `.to()` / `_apply` replaces
        every parameter's storage, anything that re-points single parameters by hand must call `_invalidate_flat`)."""
        fast = self.__dict__.get("_flat_fast")
        if fast is not None:
            flat, first, last, off_last, gfirst = fast
            if (first.data_ptr() == flat.data_ptr() and last.data_ptr() == flat.data_ptr() + 4 * off_last
                    and first.grad is not None and first.grad.data_ptr() == gfirst):
                return self._flat, self._flat_grad
        params = dict(self.named_parameters())
        dev = next(iter(params.values())).device
        if dev.type != "cuda":
            raise RuntimeError("EncoderOnlyTransformer runs on the MI355X only: move it to a cuda device "
                               "(there is no CPU path; the CPU reference lives in oracle/ for tests)")
        ok = self._flat is not None and self._flat.device == dev
        if ok:
            base = self._flat.data_ptr()
            ok = all(params[n].data_ptr() == base + 4 * off for n, (off, _) in self._layout.items())
        if not ok:
            flat = torch.zeros(self._flat_numel, dtype=torch.float32, device=dev)
            for n, (off, shape) in self._layout.items():
                view = flat[off:off + int(np.prod(shape))].view(shape)
                view.copy_(params[n].data)
                params[n].data = view
            self._flat = flat
            self._flat_grad = torch.zeros_like(flat)
            self.__dict__.pop("_scale_caches", None)     # they hold views of the old flat buffer (keyed by its address)
            self.__dict__.pop("_train_cache", None)
            self.__dict__["_param_list"] = list(params.values())
            self.__dict__["_view_cache"] = {}
            self.auto_guard.reset()                      # ... and whatever the guard measured, it measured on other weights
        gbase = self._flat_grad.data_ptr()
        for n, (off, shape) in self._layout.items():
            p = params[n]
            if p.grad is None or p.grad.data_ptr() != gbase + 4 * off:
                p.grad = self._flat_grad[off:off + int(np.prod(shape))].view(shape)
        names = list(self._layout)
        self.__dict__["_flat_fast"] = (self._flat, params[names[0]], params[names[-1]], self._layout[names[-1]][0],
                                       params[names[0]].grad.data_ptr())
        return self._flat, self._flat_grad

    def _invalidate_flat(self):
        self.__dict__.pop("_flat_fast", None)

    def load_state_dict(self, state_dict, *args, **kwargs):
        """As nn.Module.load_state_dict (the values are copied INTO the flat buffer through the parameter views); the
        AutoGuard's trust in the bound-derived scales was earned on the old weights and is withdrawn."""
        out = super().load_state_dict(state_dict, *args, **kwargs)
        self.auto_guard.reset()
        self._forget_prepared_weights()
        return out

    def weights_changed(self):
        """For callers that change parameters behind PyTorch's back (raw-pointer writes, `.data` edits, collectives on the flat
        buffer): what the AutoGuard measured and what the last optimizer step prepared (scales, bounds, planes of the
        weights) no longer describe this model.  In-place torch ops on the parameters or the flat buffer are noticed without
        this call (their version counters move: `_weights_stamp`)."""
        self.auto_guard.reset()
        self.weights_written()
        self._forget_prepared_weights()

    def weights_written(self):
        """The flat buffer was written through a raw pointer (the fused optimizer kernels, csrc/optim.hip / wprep.hip): no
        torch version counter saw it.  Moves `_weights_stamp`, so that EVERY scale cache - the evaluation pass's (dropout 0)
        beside the training pass's - prepares the weights again before its next use; the optimizer then marks the one cache
        its step prepared as fresh (`prepared_step`).  (Round 5: the stamp only knew torch's counters, and validation after
        the first epoch ran on the scales / bounds / planes of the weights of the FIRST validation.)"""
        self.__dict__["_weights_generation"] = self.__dict__.get("_weights_generation", 0) + 1

    def _forget_prepared_weights(self):
        for c in self.__dict__.get("_scale_caches", {}).values():
            c.pop("fresh", None)
        self.__dict__.pop("_train_cache", None)

    def _weights_stamp(self):
        """Moves whenever a torch op writes the flat buffer or a parameter (their views do not share one version counter) and
        whenever a fused optimizer kernel does (`weights_written`)."""
        params = self.__dict__.get("_param_list")
        if params is None:
            params = self.__dict__["_param_list"] = list(self.parameters())
        return (self._flat._version + sum(p._version for p in params), self.__dict__.get("_weights_generation", 0))

    def prepared_step(self):
        """For the fused optimizers (optim.py): (WeightsPrep plan, with_planes, mark_fresh) of the scale cache the last TRAINING
        forward pass used - the optimizer step that updates the weights then leaves their scales / bounds / planes behind for
        the next one (csrc/wprep.hip) - or None (no such pass yet, another arithmetic, feature off)."""
        tc = self.__dict__.get("_train_cache")
        if tc is None or not self.weights_prep:
            return None
        cache, with_planes = tc
        if cache.get("prep") is None:
            return None

        def mark_fresh():
            cache["fresh"] = (self._weights_stamp(), with_planes)
        return cache["prep"], with_planes, mark_fresh

    def _apply(self, fn, *a, **kw):          # .to() / .cuda() / .float(): every parameter gets a new storage
        self._invalidate_flat()
        return super()._apply(fn, *a, **kw)

    def flat_parameters(self):
        """(flat parameter buffer, flat gradient buffer): what the fused optimizer / all-reduce operate on."""
        return self._ensure_flat()

    def zero_grad(self, set_to_none=False):
        if self._flat_grad is not None:
            self._flat_grad.zero_()
        else:
            super().zero_grad(set_to_none=set_to_none)

    def grad_slices(self):
        """(offset, numel) of the flat-gradient slices in the order `_EncoderFn.backward` reports them final (the
        order of the overlapped data-parallel reduction; a rank with an empty shard walks the same list, dp.py)."""
        def span(first, last):
            o0, _ = self._layout[first]
            o1, s1 = self._layout[last]
            return o0, o1 + int(np.prod(s1)) - o0
        out = [span("output_projection.weight", "output_projection.bias")]
        for i in reversed(range(self.nlayers)):
            b = f"encoder.enc_layers.{i}."
            out.append(span(b + "self_attn.wq.weight", b + "sublayer_connections.1.norm.bias"))
        n_conv = len(self.conv_shapes or [])
        if n_conv:
            out.append(span("encoder.conv_layers.0.weight", f"encoder.conv_layers.{n_conv - 1}.bias"))
        if self.use_embedding:
            out.append(span("encoder.input_embedding.emb.weight", "encoder.input_embedding.emb.weight"))
        return out

    # ------------------------------------------------------------------ f16x2 bookkeeping (csrc/scales.hip)
    def _step_scales(self, flat, arith, p, pa, hp=True):
        """Row / column scales of every encoder weight matrix and the weight-derived bounds of the attention output, the
        FFN hidden layer and its gradient - computed on the device in three small launches per forward pass, so that no
        GEMM of the step has to run a pass over its operands for them.  Returns a list of per-layer dicts (or None when the
        arithmetic of this pass has no use for them)."""
        D, F = self.dlayer, self.dff
        if arith not in (K.GEMM_AUTO, K.GEMM_F16X2) or D > 1024 or self.nlayers == 0:
            return None
        key = (flat.data_ptr(), float(p), float(pa))
        caches = self.__dict__.setdefault("_scale_caches", {})
        if len(caches) > 4:
            caches.clear()
            self.__dict__.pop("_train_cache", None)
        cache = caches.get(key)
        if cache is None:
            dev = flat.device
            n_i = self.nlayers * (8 * D + 2 * F + 5 * 4)              # int32: weight scales + 5 uniform scales (4 copies each) per layer
            ints = torch.zeros(n_i, dtype=torch.int32, device=dev)
            stats = torch.zeros(self.nlayers, 9, 4, dtype=torch.float32, device=dev)
            factor = torch.zeros(self.nlayers, 4, dtype=torch.float32, device=dev)    # [i, 2] = dz1_factor (a stats-shaped record)
            top_stats = torch.zeros(4, dtype=torch.float32, device=dev)               # row statistics of the top layer's dy2
            ones = torch.tensor([1.0, 1.0, 1.0, 0.0], dtype=torch.float32, device=dev)     # a stats record for "no weight"
            layers, wjobs, bjobs, o = [], [], [], 0
            minbuf = torch.zeros(self.nlayers, 16, dtype=torch.int32, device=dev)   # atomicMin targets, preset before every backward pass

            def take(n):
                nonlocal o
                v = ints[o:o + n]
                o += n
                return v
            W = lambda name: self._slice(flat, name)                                     # noqa: E731
            for i in range(self.nlayers):
                b = f"encoder.enc_layers.{i}."
                wqkv, bqkv = self._qkv(flat, i)
                L = dict(rs_qkv=take(3 * D), cs_qkv=take(D), rs_o=take(D), cs_o=take(D), rs_1=take(F), cs_1=take(D),
                         rs_2=take(D), cs_2=take(F), att_scale=take(4), f1_scale=take(4), h1_scale=take(4), h2_scale=take(4),
                         dqkv_scale=take(4), dz1_factor=factor[i, 2:3], dz1_factor_rec=factor[i])
                L.update(dy2_min=minbuf[i, 0:4], dz1_min=minbuf[i, 4:8], dyo_min=minbuf[i, 8:12], dqkv_min=minbuf[i, 12:16])
                st = stats[i]
                wjobs += [dict(w=wqkv, row_scale=L["rs_qkv"], col_scale=L["cs_qkv"]),
                          dict(w=wqkv[2 * D:], stats=st[0]),                                     # W_v: row norms
                          dict(w=W(b + "self_attn.wo.weight"), row_scale=L["rs_o"], col_scale=L["cs_o"]),
                          dict(w=W(b + "pwff.layer1.weight"), row_scale=L["rs_1"], col_scale=L["cs_1"], stats=st[1]),
                          dict(w=W(b + "pwff.layer2.weight"), row_scale=L["rs_2"], col_scale=L["cs_2"], stats=st[2]),
                          dict(w=W(b + "sublayer_connections.0.norm.weight"), stats=st[3]),
                          dict(w=W(b + "sublayer_connections.0.norm.bias"), stats=st[4]),
                          dict(w=W(b + "sublayer_connections.1.norm.weight"), stats=st[5]),
                          dict(w=W(b + "sublayer_connections.1.norm.bias"), stats=st[6]),
                          dict(w=bqkv[2 * D:], stats=st[7]),                                     # b_v
                          dict(w=W(b + "pwff.layer1.bias"), stats=st[8])]
                sq = math.sqrt(D)
                bjobs += [dict(ln_gamma=st[3], ln_beta=st[4], w=st[0], w_index=0, bias=st[7], sqrt_d=sq,
                               post_scale=1.0 / (1.0 - pa), out_scale=L["att_scale"]),
                          dict(ln_gamma=st[5], ln_beta=st[6], w=st[1], w_index=0, bias=st[8], sqrt_d=sq,
                               post_scale=1.0 / (1.0 - p), out_scale=L["f1_scale"]),
                          dict(w=st[2], w_index=1, post_scale=1.0 / (1.0 - p), out_value=L["dz1_factor"]),
                          # |LN(x)|_inf <= |LN(x)|_2 <= max|gamma| sqrt(D) + |beta|_2: ONE scale for h1 / h2 as operands of dW
                          dict(ln_gamma=st[3], ln_beta=st[4], w=ones, w_index=0, sqrt_d=sq, out_scale=L["h1_scale"]),
                          dict(ln_gamma=st[5], ln_beta=st[6], w=ones, w_index=0, sqrt_d=sq, out_scale=L["h2_scale"])]
                layers.append(L)
            assert o == n_i
            dq_stats = torch.zeros(self.nlayers, 4, dtype=torch.float32, device=dev)
            gstats = torch.zeros(self.nlayers, len(AutoGuard.SITES), 4, dtype=torch.float32, device=dev)   # AutoGuard measurements
            for i, L in enumerate(layers):
                L["dqkv_stats"] = dq_stats[i]
                L["minbuf"] = minbuf
                L["guard_stats"], L["ints"] = gstats, ints
                L["top_stats"] = top_stats
            cache = caches[key] = dict(layers=layers, wjobs=wjobs, bjobs=bjobs,
                                       keep=(ints, stats, factor, ones, dq_stats, minbuf, gstats, top_stats))
            # the FFN-layer-1 weights (behind the second LayerNorm), pre-split once per forward pass for ptamd_gemm_hp.  (The QKV
            # weights were too until the producers of the staging GEMM were trimmed at the end of round 3: QKV now runs
            # 34.7 / 58.5 / 106.8 us on it at 4096 / 8192 / 16384 tokens against 35.4 / 70.5 / 110.3 on ptamd_gemm_hp, and the
            # first LayerNorm no longer writes planes; FFN-1, with its ReLU + dropout epilogue, stays: 142 against 167 us.)
            cache["hp_mats"], cache["hp_outs"] = [], []
            cache["hpT_mats"], cache["hpT_scales"], cache["hpT_outs"] = [], [], []
            if D % 32 == 0 and D <= 2048:         # (built whatever `hp_forward` says now: the flag is read at use time)
                for i, L in enumerate(layers):
                    w1 = W(f"encoder.enc_layers.{i}.pwff.layer1.weight")
                    # (the planes' row scales ARE the rows' f16x2 scales: one array serves the staging GEMM and ptamd_gemm_hp)
                    L["hp_1"] = K.hp_view(torch.empty(K.lib().ptamd_hp_bytes(F, D), dtype=torch.uint8, device=dev), L["rs_1"], F, D)
                    L["hp_qkv"] = K.hp_view(torch.empty(K.lib().ptamd_hp_bytes(3 * D, D), dtype=torch.uint8, device=dev),
                                            L["rs_qkv"], 3 * D, D)
                    cache["hp_mats"] += [w1, self._qkv(flat, i)[0]]
                    cache["hp_outs"] += [L["hp_1"], L["hp_qkv"]]
                    # W2^T as the B operand of dX = dy2 W2 (operand rows = the F columns of W2 [D, F], contraction over D): split
                    # with the column scales computed above
                    w2 = W(f"encoder.enc_layers.{i}.pwff.layer2.weight")
                    planes2 = torch.empty(K.lib().ptamd_hp_bytes(F, D), dtype=torch.uint8, device=dev)
                    L["hp_2t"] = K.hp_view(planes2, L["cs_2"], F, D)
                    cache["hpT_mats"] += [w2]
                    cache["hpT_scales"] += [L["cs_2"]]
                    cache["hpT_outs"] += [L["hp_2t"]]
            # (csrc/wprep.hip keeps whole rows of <= 512 columns in registers and walks wider matrices as 512-column panels:
            # widths above 512 that are not multiples of 512 - d_model 768, d_ff 1000 - take the separate launches below)
            panels_ok = all(c <= 512 or c % 512 == 0 for c in (D, F))
            cache["prep"] = self._build_prep(cache, p, pa) if (F <= 2048 and D % 4 == 0 and F % 4 == 0 and panels_ok) else None
        need_planes = bool(cache["hp_mats"]) and bool(hp)
        if self.weights_prep and cache.get("prep") is not None:
            # ONE pass over the weights (two launches) - or none at all when the optimizer step that wrote these weights left
            # everything behind (`prepared_step`) and nothing has touched them since
            stamp = self._weights_stamp()
            fresh = cache.get("fresh")
            if fresh is None or fresh[0] != stamp or (need_planes and not fresh[1]):
                cache["prep"].prepare(self._flat, with_planes=need_planes)
                cache["fresh"] = (stamp, need_planes)
                self.__dict__["_prep_launches"] = self.__dict__.get("_prep_launches", 0) + 1
            if self.training and self.__dict__.get("_fwd_grad", False):
                self.__dict__["_train_cache"] = (cache, need_planes)
            return cache["layers"]
        K.weight_scales(cache["wjobs"])
        K.bound_scales(cache["bjobs"])
        if cache["hp_mats"] and hp:
            # (only what this pass will read: the QKV planes with `hp_qkv`, W2^T - an operand of the backward pass - when one follows)
            if self.hp_qkv:
                K.hp_split_rows(cache["hp_mats"], cache["hp_outs"])
            else:
                K.hp_split_rows(cache["hp_mats"][0::2], cache["hp_outs"][0::2])
            if self.hp_dx and F % 32 == 0 and self.__dict__.get("_fwd_grad", False):
                K.hp_split_cols(cache["hpT_mats"], cache["hpT_scales"], cache["hpT_outs"])
        return cache["layers"]

    def _build_prep(self, cache, p, pa):
        """The description of this model's encoder weights for csrc/wprep.hip: the matrices / vectors of `_step_scales`' job
        lists as disjoint segments of the flat buffer (W_v's statistics are rows 2D.. of the W_qkv segment), the bound jobs per
        layer with statistics records instead of pointers."""
        D, F = self.dlayer, self.dff
        layers = cache["layers"]
        ints, factor = cache["keep"][0], cache["keep"][2]
        off = lambda name: self._layout[name][0]                                              # noqa: E731
        planes_ok = bool(cache["hp_mats"])
        segs, groups = [], []
        for i, L in enumerate(layers):
            b = f"encoder.enc_layers.{i}."
            r = lambda k, i=i: 9 * i + k                                                       # noqa: E731  (statistics record)
            segs += [
                dict(offset=off(b + "self_attn.wq.weight"), rows=3 * D, cols=D, row_scale=L["rs_qkv"], col_scale=L["cs_qkv"],
                     stats=r(0), stats_row0=2 * D, row_planes=L["hp_qkv"].planes if planes_ok else None),
                dict(offset=off(b + "self_attn.wq.bias") + 2 * D, rows=1, cols=D, stats=r(7)),             # b_v
                dict(offset=off(b + "self_attn.wo.weight"), rows=D, cols=D, row_scale=L["rs_o"], col_scale=L["cs_o"]),
                dict(offset=off(b + "pwff.layer1.weight"), rows=F, cols=D, row_scale=L["rs_1"], col_scale=L["cs_1"], stats=r(1),
                     row_planes=L["hp_1"].planes if planes_ok else None),
                dict(offset=off(b + "pwff.layer1.bias"), rows=1, cols=F, stats=r(8)),
                dict(offset=off(b + "pwff.layer2.weight"), rows=D, cols=F, row_scale=L["rs_2"], col_scale=L["cs_2"], stats=r(2),
                     colnorm=True, col_planes=L["hp_2t"].planes if (planes_ok and F % 32 == 0) else None),
                dict(offset=off(b + "sublayer_connections.0.norm.weight"), rows=1, cols=D, stats=r(3)),
                dict(offset=off(b + "sublayer_connections.0.norm.bias"), rows=1, cols=D, stats=r(4)),
                dict(offset=off(b + "sublayer_connections.1.norm.weight"), rows=1, cols=D, stats=r(5)),
                dict(offset=off(b + "sublayer_connections.1.norm.bias"), rows=1, cols=D, stats=r(6))]
            sq = math.sqrt(D)
            groups.append([
                dict(ln_gamma=r(3), ln_beta=r(4), w=r(0), w_index=0, bias=r(7), sqrt_d=sq, post_scale=1.0 / (1.0 - pa),
                     out_scale=L["att_scale"]),
                dict(ln_gamma=r(5), ln_beta=r(6), w=r(1), w_index=0, bias=r(8), sqrt_d=sq, post_scale=1.0 / (1.0 - p),
                     out_scale=L["f1_scale"]),
                dict(w=r(2), w_index=1, post_scale=1.0 / (1.0 - p), out_value=L["dz1_factor"]),
                dict(ln_gamma=r(3), ln_beta=r(4), w=-1, w_index=0, sqrt_d=sq, out_scale=L["h1_scale"]),
                dict(ln_gamma=r(5), ln_beta=r(6), w=-1, w_index=0, sqrt_d=sq, out_scale=L["h2_scale"])])
        return K.WeightsPrep(self._flat_numel, segs, groups, ints, factor.view(-1), 9 * self.nlayers)

    def _slice(self, buf, name):
        """View of parameter `name` in a flat buffer (the parameters or the gradients); the views of the two long-lived
        buffers are cached (46 view constructions per step were 5 % of the host time of a launch-bound step)."""
        cache = self.__dict__.setdefault("_view_cache", {})
        ptr = buf.data_ptr()
        key = (ptr, name)
        v = cache.get(key)
        if v is None:
            off, shape = self._layout[name]
            v = buf[off:off + int(np.prod(shape))].view(shape)
            # (the autograd anchor of a forward pass is a detached alias of the flat buffer: same address, same views)
            if (self._flat is not None and ptr == self._flat.data_ptr()) or \
                    (self._flat_grad is not None and ptr == self._flat_grad.data_ptr()):
                cache[key] = v.detach()
                v = cache[key]
        return v

    def _qkv(self, buf, i):
        cache = self.__dict__.setdefault("_view_cache", {})
        key = (buf.data_ptr(), "qkv", i)
        v = cache.get(key)
        if v is None:
            b = f"encoder.enc_layers.{i}.self_attn."
            off_w, _ = self._layout[b + "wq.weight"]
            off_b, _ = self._layout[b + "wq.bias"]
            D = self.dlayer
            v = (buf[off_w:off_w + 3 * D * D].view(3 * D, D).detach(), buf[off_b:off_b + 3 * D].detach())
            if (self._flat is not None and key[0] == self._flat.data_ptr()) or \
                    (self._flat_grad is not None and key[0] == self._flat_grad.data_ptr()):
                cache[key] = v
        return v

    # ------------------------------------------------------------------ forward
    def forward(self, enc_input, dec_input=None):
        flat, _ = self._ensure_flat()
        if enc_input.shape[1] > self.max_seq_len:
            raise RuntimeError(f"sequence length {enc_input.shape[1]} exceeds max_seq_len={self.max_seq_len} "
                               "(the reference's positional table has the same limit, Sublayers.py:48,60)")
        seq = enc_input.to(flat.device, torch.int64).contiguous()
        if self.training:
            self._step_counter += 1
        seed = (self.dropout_seed + 0x9E3779B97F4A7C15 * self._step_counter) & (2 ** 63 - 1)
        # the flat buffer is a leaf of the autograd graph only so that backward() reaches _EncoderFn.backward;
        # gradients are written into the flat gradient buffer directly
        anchor = flat.detach().requires_grad_(torch.is_grad_enabled())
        self.__dict__["_fwd_grad"] = torch.is_grad_enabled()      # (grad mode is off inside autograd.Function.forward)
        out = _EncoderFn.apply(anchor, seq, self, seed)
        return out.view(seq.shape[0], seq.shape[1], NUM_PREDICTED_ANGLES * 2)

    def predict(self, enc_input):
        return self.forward(enc_input)

    def set_dropout(self, p, attn_p=None):
        """Set every dropout of the model (parity runs use 0 everywhere)."""
        self.dropout = float(p)
        self.attn_dropout = float(p if attn_p is None else attn_p)


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat, seq, model, seed):
        m = model
        B, L = seq.shape
        D, H = m.dlayer, m.nhead
        train = m.training
        p = m.dropout if train else 0.0
        pa = m.attn_dropout if train else 0.0
        W = lambda name: m._slice(flat, name)                                      # noqa: E731
        ar = K.get_gemm_mode() if m.gemm_mode is None else int(m.gemm_mode)        # every launch below carries it
        attn_default = ar                 # the attention kernels find their f16x2 scales themselves: AUTO stays AUTO for them
        if ar == K.GEMM_AUTO and B * L * D < AUTO_F16X2_MIN_WORK:
            ar = K.GEMM_BF16X3         # launch-bound step: the scale bookkeeping of f16x2 costs more than its products save
        pe = m.encoder.positional_enc.pe[0]
        # ---- front end: embedding (+ doubled positional add) or one-hot, then the optional Conv1d stack
        if m.use_embedding:
            x = K.embed_fwd(seq, W("encoder.input_embedding.emb.weight"), pe, p, seed)
            cin = m.dmodel
        else:
            cin = len(m.vocab)
            x = K.onehot(seq, cin)
        conv_saved = []
        for j, (ci, co, k) in enumerate(m.conv_shapes or []):
            y, w2 = K.conv1d_fwd(x, B, L, ci, W(f"encoder.conv_layers.{j}.weight"), W(f"encoder.conv_layers.{j}.bias"), k,
                                   arith=ar)
            conv_saved.append((x, w2))
            x, cin = y, co
        if not m.use_embedding:
            x = K.posenc_add_fwd(x, pe, B, L, p, seed)             # enc_output += positional_enc(enc_output)
        saved = []
        # f16x2 row scales without passes over the operands: weights and weight-derived bounds once per forward, LayerNorm
        # outputs from the LayerNorm kernel itself (None: the arithmetic of this pass does not use them)
        Tn = B * L
        # (below HP_MIN_TOKENS the staging GEMM with 128-row tiles is the faster of the two: profiles/r03/r03_tile_height.txt)
        want_hp = Tn >= HP_MIN_TOKENS and bool(m.hp_forward)
        scales = m._step_scales(flat, ar, p, pa, hp=want_hp)
        # QKV and FFN layer 1 (the products behind the two LayerNorms) run on ptamd_gemm_hp: the LayerNorm kernel writes its
        # output a second time as pre-split planes (one buffer, consumed at once), the weights were split above.
        use_hp = want_hp and scales is not None and "hp_1" in scales[0]
        hplanes = torch.empty(K.lib().ptamd_hp_bytes(Tn, D), dtype=torch.uint8, device=x.device) if use_hp else None
        # the guard of the bound-derived scales: sites whose measured slack is too large do not use their bound (AutoGuard)
        guard = m.auto_guard if (scales is not None and m.auto_guard.enabled) else None
        measure = False
        if guard is not None:
            guard.poll()
            if train and m.__dict__.get("_fwd_grad", False):          # a training step: a backward pass will follow
                measure = guard.want_measure()
                guard.count_step()
        # a pass that no backward pass follows (evaluation, inference) measures in the forward pass: the four forward operands
        # (dz1 exists in backward only) - the first such pass on these weights, so that inference does not stay on the
        # untrusting path, and every `interval`-th one after it (other inputs, same weights: the slack is a property of both)
        measure_fwd = False
        if guard is not None and not (train and m.__dict__.get("_fwd_grad", False)):
            measure_fwd = guard.want_measure_forward()
            guard.eval_passes += 1
        off = guard.off if guard is not None else None
        wide = guard.wide if guard is not None else None
        ctx_off = None if off is None else (off.copy(), wide.copy())

        def prod(i, k, **scale_kw):
            """arithmetic + scales of activation x weight product k of layer i (AutoGuard.WIDE order): bf16x3 without scales
            when the weight's scales spread too far along the contracted index, else the pass's arithmetic with `scale_kw`"""
            if wide is not None and wide[i, k]:
                return dict(arith=K.GEMM_BF16X3)
            return dict(arith=ar, **scale_kw)
        for i in range(m.nlayers):
            b = f"encoder.enc_layers.{i}."
            sid = i * 8
            sc = scales[i] if scales is not None else None
            wqkv, bqkv = m._qkv(flat, i)
            s_h1 = torch.empty(Tn, dtype=torch.int32, device=x.device) if sc else None
            hp_q = use_hp and m.hp_qkv and not (wide is not None and wide[i, 0])
            h1, mean1, rstd1 = K.layernorm_fwd(x, W(b + "sublayer_connections.0.norm.weight"),
                                               W(b + "sublayer_connections.0.norm.bias"), row_scale=s_h1,
                                               planes=hplanes if hp_q else None)
            if isinstance(x, K.PendingRows):          # (the layer below left its output as K slices: made by this LayerNorm)
                x = x.value
            attn_ar_f = attn_default if m.attn_mode is None else m.attn_mode
            # K and V leave the QKV product as the pre-split planes the attention kernels of this batch shape read (no fp32 K / V at
            # all: the forward kernel fills its stages by LDS-DMA, the one-sweep backward kernel loads its key rows from them)
            kv = None
            if hp_q and m.kv_planes and K.attention_reads_kv_planes(B, L, H, D // H, attn_ar_f):
                kv = K.attention_kv_buffers(Tn, H, x.device)
                m.__dict__["_kv_plane_passes"] = m.__dict__.get("_kv_plane_passes", 0) + 1
            if hp_q:
                qkv = K.gemm_hp(K.hp_view(hplanes, s_h1, Tn, D), sc["hp_qkv"],
                                torch.empty(Tn, 3 * D, dtype=torch.float32, device=x.device), bias=bqkv, kv=kv, kv_col0=D, kv_heads=H)
            else:
                qkv = K.linear_fwd(h1, wqkv, bqkv, **prod(i, 0, a_scale=s_h1, b_scale=sc and sc["rs_qkv"]))
            # the dropout decisions of the probabilities, kept for the backward kernel that reads them instead of drawing
            # them again (8 MB per layer at 32 x 512 x 8 heads; only where that kernel will run)
            kbits = None
            if pa > 0.0 and m.keep_attn_bits and m.__dict__.get("_fwd_grad", False) and \
                    K.attention_bwd_reads_keep_bits(B, L, H, D // H, attn_ar_f):
                kbits = K.attention_keep_bits(B, L, H, x.device)
                m.__dict__["_attn_bits_passes"] = m.__dict__.get("_attn_bits_passes", 0) + 1
            att, lse = K.attention_fwd(qkv, seq, H, pa, seed, sid + _SITE_ATTN, arith=attn_ar_f, keep_bits=kbits, kv=kv)
            use_b = sc is not None and not (off is not None and off[i, 0])          # att on its bound (else: exact row scales)
            x2 = K.linear_fwd(att, W(b + "self_attn.wo.weight"), W(b + "self_attn.wo.bias"), residual=x, ldr=D,
                              dropout_p=p, seed=seed, stream_id=sid + _SITE_ATTN_OUT,
                              **prod(i, 1, a_scale=sc["att_scale"] if use_b else None, a_scale_stride=0 if use_b else 1,
                                     b_scale=sc and sc["rs_o"]))
            s_h2 = torch.empty(Tn, dtype=torch.int32, device=x.device) if sc else None
            h2, mean2, rstd2 = K.layernorm_fwd(x2, W(b + "sublayer_connections.1.norm.weight"),
                                               W(b + "sublayer_connections.1.norm.bias"), row_scale=s_h2, planes=hplanes)
            fmask = None
            if use_hp and not (wide is not None and wide[i, 2]):
                # (with a backward pass to come: the epilogue also leaves "f1 > 0" as one bit per element - what the gated dX
                # product of layer 2 reads instead of the 134 MB of f1)
                if m.ffn_gate_mask and m.__dict__.get("_fwd_grad", False):
                    fmask = K.gate_mask_buffer(Tn, m.dff, x.device)
                    m.__dict__["_gate_mask_passes"] = m.__dict__.get("_gate_mask_passes", 0) + 1
                f1 = K.gemm_hp(K.hp_view(hplanes, s_h2, Tn, D), sc["hp_1"],
                               torch.empty(Tn, m.dff, dtype=torch.float32, device=x.device), bias=W(b + "pwff.layer1.bias"),
                               flags=K.EPI_RELU, dropout_p=p, seed=seed, stream_id=sid + _SITE_FFN_HID, gate_mask_out=fmask)
            else:
                f1 = K.linear_fwd(h2, W(b + "pwff.layer1.weight"), W(b + "pwff.layer1.bias"), flags=K.EPI_RELU,
                                  dropout_p=p, seed=seed, stream_id=sid + _SITE_FFN_HID,
                                  **prod(i, 2, a_scale=s_h2, b_scale=sc and sc["rs_1"]))
            use_b = sc is not None and not (off is not None and off[i, 1])          # f1 on its bound
            # (few tokens: the K slices of this product go to the next layer's LayerNorm unreduced - kernels.PendingRows)
            x3 = K.linear_fwd(f1, W(b + "pwff.layer2.weight"), W(b + "pwff.layer2.bias"), residual=x2, ldr=D,
                              dropout_p=p, seed=seed, stream_id=sid + _SITE_FFN_OUT, defer_reduce=i + 1 < m.nlayers,
                              **prod(i, 3, a_scale=sc["f1_scale"] if use_b else None, a_scale_stride=0 if use_b else 1,
                                     b_scale=sc and sc["rs_2"]))
            saved.append((x, mean1, rstd1, h1, qkv, att, lse, x2, mean2, rstd2, h2, f1, kbits, fmask, kv))
            if measure_fwd:
                gs = sc["guard_stats"][i]
                K.weight_scales([dict(w=t, stats=gs[j], rows_only=True) for j, t in ((0, att), (1, f1), (3, h1), (4, h2))])
            x = x3
        pred = K.linear_fwd(x, W("output_projection.weight"), W("output_projection.bias"),
                            flags=K.EPI_TANH if m.use_tanh_out else 0, arith=ar)
        if measure_fwd:
            scales[0]["guard_stats"][:, 2].zero_()                 # dz1: not measured here (slack 0 = "as good as its bound")
            K.fill_u32([(scales[0]["minbuf"], 0x7F000000)])        # ... and its scale slot reads "unused"
            guard.submit(scales[0]["guard_stats"], scales[0]["ints"], scales[0]["minbuf"], scales, forward_only=True)
        ctx.model, ctx.seed, ctx.seq, ctx.flat, ctx.arith, ctx.attn_arith = m, seed, seq, flat, ar, attn_default
        ctx.p, ctx.pa = p, pa
        ctx.guard, ctx.measure, ctx.off = guard, measure, ctx_off
        ctx.use_hp = use_hp
        ctx.saved, ctx.conv_saved, ctx.scales = saved, conv_saved, scales
        # (NOT `ctx.pred = pred`: the output's grad_fn is this node, the node would hold the output - a reference cycle only the
        # cyclic collector frees, at a time of its choosing, with everything else the node still holds.  A detached alias of the
        # same storage carries no grad_fn.)
        ctx.x_last, ctx.pred = x, pred.detach()
        return pred

    @staticmethod
    def backward(ctx, dpred):
        m, seed, seq, flat = ctx.model, ctx.seed, ctx.seq, ctx.flat
        p, pa, ar = ctx.p, ctx.pa, ctx.arith
        B, L = seq.shape
        D, H = m.dlayer, m.nhead
        gflat = m._flat_grad
        m.__dict__["_grad_dirty"] = True          # raw-pointer writes into the flat gradient buffer from here on (optim.py: zero_grad)
        W = lambda name: m._slice(flat, name)                                      # noqa: E731
        G = lambda name: m._slice(gflat, name)                                     # noqa: E731

        ln_pending = []      # deferred (dgamma, dbeta) reductions of the LayerNorm backward kernels: one launch per flush
        # Small batches: the weight-gradient products (independent of the dX chain, 40 % of the backward GEMM time) go to a
        # side stream; the main stream waits for it before a gradient slice is handed on and at the end of the pass.
        main = torch.cuda.current_stream(dpred.device)
        side = None
        if m.side_stream_dw and D >= 512 and B * L >= SIDE_STREAM_MIN_TOKENS and B * L * D <= SIDE_STREAM_MAX_WORK:
            side = m.__dict__.get("_side_stream")
            if side is None or side.device != dpred.device:
                side = m.__dict__["_side_stream"] = torch.cuda.Stream(device=dpred.device)

        # Operands the side stream reads were allocated on the main stream.  They are kept ALIVE here until the main stream has
        # waited for the side stream (`join`, the end of the pass) instead of being handed to `record_stream`: blocks parked
        # behind a side-stream event return to the allocator whenever the host happens to notice - the free lists of one step
        # differed from the next one's and reserved memory crept (9.3 -> 9.9 GiB over 4,000 steps, profiles/r05/r05_soak.txt);
        # with plain lifetimes every step allocates and frees in the same order.  (~3 GB more at the peak of 32 x 512.)
        side_keep = []

        def dw(*tensors_then_kwargs, **kw):
            """K.linear_bwd_weight(dy, x, dw, db, ...) - on the side stream when there is one."""
            if side is None:
                return K.linear_bwd_weight(*tensors_then_kwargs, **kw)
            side.wait_stream(main)                       # the operands were produced on the main stream
            side_keep.extend(tensors_then_kwargs[:2])    # allocated on the main stream, read on the side stream
            with torch.cuda.stream(side):
                return K.linear_bwd_weight(*tensors_then_kwargs, **kw)

        # The weight-gradient products of a layer go out in GROUPS when they run in f16x2 on uniform scales (ptamd_gemm_group:
        # one launch whose work items fill the chip in whole rounds with a longer K per item and fewer slabs, one launch for all
        # the split-K reductions) - their operands are only read afterwards.  m.dw_group: "pairs" = the two FFN products
        # when both exist, the two attention products at the end of the layer (best where a pair fills a round on its own:
        # 10.39 against 10.57 ms at 32 x 512, the whole layer as one group 10.63 - it starts late and runs three rounds);
        # "layer" = all four at the end (best for few tokens, where the step is bound by launches: config 3 7.31 / 7.44 / 7.65
        # ms as layer / pairs / one by one); "auto" picks by the size of the FFN pair; "off" = one by one.
        queue = []
        dw_group = m.dw_group if K.GROUP_DW else "off"
        if dw_group == "auto":
            ffn_tiles = -(-D // 256) * -(-m.dff // 128) + -(-m.dff // 256) * -(-D // 128)      # 256 x 128 tiles of dW2 and dW1
            dw_group = "pairs" if ffn_tiles * B * L >= 256 * 2048 else "layer"

        def dw_later(dy, x, gw, gb, arith=None, dy_scale=None, x_scale=None):
            if dw_group == "off":
                return dw(dy, x, gw, gb, arith=arith, dy_scale=dy_scale, x_scale=x_scale)
            queue.append((dy, x, gw, gb, arith, dy_scale, x_scale))

        def dw_flush():
            jobs = list(queue)
            queue.clear()
            # members of a group: f16x2 on uniform scales (the top layer's FFN gradients come without - no fused LayerNorm
            # backward above them - and go one by one, like everything in the other arithmetics)
            ok = [j for j in jobs if j[5] is not None and j[6] is not None and j[1].shape[1] % 4 == 0]
            tiles = sum(-(-j[0].shape[1] // 256) * -(-j[1].shape[1] // 128) for j in ok)
            sk = K.pick_group_split(tiles, ok[0][0].shape[0]) if 2 <= len(ok) <= 4 else 0
            if sk < 2:
                ok = []
            for dy, x, gw, gb, a, sy, sx in jobs:
                if not any(dy is o[0] and x is o[1] for o in ok):
                    dw(dy, x, gw, gb, arith=a, dy_scale=sy, x_scale=sx)
            if not ok:
                return
            group = [(dy, x, gw, gb, sy, sx) for dy, x, gw, gb, a, sy, sx in ok]
            if side is None:
                return K.linear_bwd_weight_group(group, sk)
            side.wait_stream(main)
            for dy, x, *_ in ok:
                side_keep.extend((dy, x))
            with torch.cuda.stream(side):
                K.linear_bwd_weight_group(group, sk)

        def join():
            if side is not None:
                main.wait_stream(side)
                side_keep.clear()                # (whatever is freed from here on is reused behind the side stream's work)

        def done(first, last):
            if m.grad_hook is not None:
                K.layernorm_bwd_flush(ln_pending)     # a slice handed to the all-reduce must be final
                o0, _ = m._layout[first]
                o1, s1 = m._layout[last]
                if side is None:
                    m.grad_hook(o0, o1 + int(np.prod(s1)) - o0)
                else:
                    # the slice is final once BOTH streams are here: the collective is issued from the side stream behind a
                    # wait for the main one, so the dX chain on the main stream does not stop for the weight-gradient products
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        m.grad_hook(o0, o1 + int(np.prod(s1)) - o0)

        off, wide = ctx.off if ctx.off is not None else (None, None)

        def prod(i, k, **scale_kw):      # as in forward: dX products 4..7 of AutoGuard.WIDE
            if wide is not None and wide[i, k]:
                return dict(arith=K.GEMM_BF16X3)
            return dict(arith=ar, **scale_kw)
        dpred = dpred.contiguous().view(-1, NUM_PREDICTED_ANGLES * 2)
        dpre = K.tanh_bwd(dpred, ctx.pred) if m.use_tanh_out else dpred
        K.linear_bwd_weight(dpre, ctx.x_last, G("output_projection.weight"), G("output_projection.bias"),
                            arith=ar)
        dx = K.linear_bwd_input(dpre, W("output_projection.weight"), arith=ar)
        done("output_projection.weight", "output_projection.bias")
        scales = ctx.scales
        fuse = D <= 1024                                   # LayerNorm backward + the dropout backward behind it in one kernel
        i32 = lambda: torch.empty(B * L, dtype=torch.int32, device=dx.device)                 # noqa: E731
        dy2 = s_dy2 = bs_dz1 = None                        # dropout'(dx) of the FFN output site, made by the layer above
        attn_ar = ctx.attn_arith if m.attn_mode is None else int(m.attn_mode)
        s_dqkv_all = None
        have_min = False                                   # ... together with the uniform scales of dy2 / dz1 for the dW products
        if scales is not None:
            # atomicMin targets of this backward pass (largest scale): the uniform scales of every layer and, where the f16x2
            # attention kernels leave the row scales of dqkv behind, those - one launch (ptamd_fill_u32) for both presets
            presets = [(scales[0]["minbuf"], 0x7F000000)]
            if m.attn_row_scales and K.attention_row_scales_available(D // H, attn_ar):
                s_dqkv_all = torch.empty((m.nlayers, B * L), dtype=torch.int32, device=dpred.device)
                presets.append((s_dqkv_all, 0x7F000000))
            K.fill_u32(presets)
        # dX of FFN layer 2 on ptamd_gemm_hp: the fused LayerNorm backward that makes dy2 writes it a second time as planes
        dy2_planes, have_planes = None, False
        if ctx.use_hp and m.hp_dx and fuse and scales is not None and "hp_2t" in scales[0] and m.dff % 32 == 0:
            dy2_planes = torch.empty(K.lib().ptamd_hp_bytes(B * L, D), dtype=torch.uint8, device=dx.device)
        for i in reversed(range(m.nlayers)):
            b = f"encoder.enc_layers.{i}."
            sid = i * 8
            sc = scales[i] if scales is not None else None
            x, mean1, rstd1, h1, qkv, att, lse, x2, mean2, rstd2, h2, f1, kbits, fmask, kv = ctx.saved[i]
            # x3 = x2 + drop(f1 W2^T + b2)
            if dy2 is None:
                dy2 = K.dropout_bwd(dx, p, seed, sid + _SITE_FFN_OUT) if p > 0 else dx
                if sc is not None and fuse and m.top_layer_scales:
                    # the top layer: no fused LayerNorm backward above it that would leave the uniform scales of dy2 / dz1
                    # behind - one streaming pass over dy2 (largest |x| and largest row norm) and the same bound instead,
                    # so that its two FFN weight-gradient products run in f16x2 like those of every other layer
                    ts = sc["top_stats"]
                    K.weight_scales([dict(w=dy2, stats=ts, rows_only=True)])
                    K.bound_scales([dict(w=ts, w_index=2, out_scale=sc["dy2_min"]),
                                    dict(ln_gamma=sc["dz1_factor_rec"], sqrt_d=1.0, w=ts, w_index=0, out_scale=sc["dz1_min"])])
                    have_min = True
            uni = sc is not None and have_min              # uniform scales of both operands: the dW product runs in f16x2
            o_att, o_f1, o_dz1, o_h1, o_h2 = (bool(v) for v in off[i]) if off is not None else (False,) * 5   # AutoGuard
            dw_later(dy2, f1, G(b + "pwff.layer2.weight"), G(b + "pwff.layer2.bias"), arith=ar,
                                dy_scale=sc["dy2_min"] if uni and not o_f1 else None, x_scale=sc["f1_scale"] if uni and not o_f1 else None)
            # backward of layer2 and, in its epilogue, of the ReLU + dropout in front of it (gate = saved f1)
            if dy2_planes is not None and have_planes and not (wide is not None and wide[i, 7]):
                # dy2 came from the fused LayerNorm backward of the layer above together with its planes: LDS-DMA kernel
                dz1 = K.gemm_hp(K.hp_view(dy2_planes, s_dy2, B * L, D), sc["hp_2t"],
                                torch.empty(B * L, m.dff, dtype=torch.float32, device=dx.device),
                                residual=f1 if fmask is None else None, ldr=f1.stride(0) if fmask is None else 0,
                                gate_mask=fmask, flags=K.EPI_GATE, gate_scale=1.0 / (1.0 - p))
            else:
                dz1 = K.linear_bwd_input(dy2, W(b + "pwff.layer2.weight"), gate=f1, gate_dropout_p=p, gate_mask=fmask,
                                         **prod(i, 7, a_scale=s_dy2 if sc else None, b_scale=sc and sc["cs_2"]))
            if ctx.measure:     # the true maxima of the five bound-scaled operands of this layer (AutoGuard; every 16th step)
                gs = sc["guard_stats"][i]
                K.weight_scales([dict(w=t, stats=gs[j], rows_only=True) for j, t in enumerate((att, f1, dz1, h1, h2))])
            uni1 = uni and not (o_dz1 or o_h2)
            dw_later(dz1, h2, G(b + "pwff.layer1.weight"), G(b + "pwff.layer1.bias"), arith=ar,
                                dy_scale=sc["dz1_min"] if uni1 else None, x_scale=sc["h2_scale"] if uni1 else None)
            if dw_group != "layer":
                dw_flush()                                    # the two FFN products now, the two attention products at the end
            # (few tokens: the K slices of dh2 / dh1 go to the fused LayerNorm backward unreduced - kernels.Slabs)
            dh2 = K.linear_bwd_input(dz1, W(b + "pwff.layer1.weight"), defer_reduce=fuse,
                                     **prod(i, 6, a_scale=bs_dz1 if sc and not o_dz1 else None, b_scale=sc and sc["cs_1"]))
            # x2 = x + drop(att Wo^T + bo)
            g2w, g2b = G(b + "sublayer_connections.1.norm.weight"), G(b + "sublayer_connections.1.norm.bias")
            if fuse:
                s_dyo = i32() if sc else None
                dx2, dyo = K.layernorm_bwd_dropout(dh2, x2, W(b + "sublayer_connections.1.norm.weight"), mean2, rstd2, g2w, g2b,
                                                   dx, p, seed, sid + _SITE_ATTN_OUT, row_scale=s_dyo,
                                                   row_scale_min=sc["dyo_min"] if sc else None, pending=ln_pending)
            else:
                s_dyo = None
                dx2 = K.layernorm_bwd(dh2, x2, W(b + "sublayer_connections.1.norm.weight"), mean2, rstd2, g2w, g2b, dres=dx,
                                      pending=ln_pending)
                dyo = K.dropout_bwd(dx2, p, seed, sid + _SITE_ATTN_OUT) if p > 0 else dx2
            uni_o = sc is not None and fuse and not o_att
            dw_later(dyo, att, G(b + "self_attn.wo.weight"), G(b + "self_attn.wo.bias"), arith=ar,
                                dy_scale=sc["dyo_min"] if uni_o else None, x_scale=sc["att_scale"] if uni_o else None)
            datt = K.linear_bwd_input(dyo, W(b + "self_attn.wo.weight"), **prod(i, 5, a_scale=s_dyo, b_scale=sc and sc["cs_o"]))
            # the f16x2 attention kernels leave the row scales of dqkv (A of the dX product) and the smallest of them (the
            # uniform scale of dqkv as operand of the dW product) behind; other arithmetics: one pass over dqkv
            attn_scales = sc is not None and m.attn_row_scales and K.attention_row_scales_available(D // H, attn_ar)
            s_dqkv = s_dqkv_all[i] if attn_scales else None
            dqkv = K.attention_bwd(qkv, seq, att, datt, lse, H, pa, seed, sid + _SITE_ATTN, arith=attn_ar,
                                    row_scale=s_dqkv, row_scale_min=sc["dqkv_min"] if attn_scales else None, keep_bits=kbits,
                                    kv=kv)
            gw, gb = m._qkv(gflat, i)
            wqkv, _ = m._qkv(flat, i)
            if sc is not None:
                dq_uni = sc["dqkv_min"]
                if not attn_scales:
                    s_dqkv, dq_uni = i32(), sc["dqkv_scale"]
                    K.weight_scales([dict(w=dqkv, row_scale=s_dqkv, stats=sc["dqkv_stats"], rows_only=True)])
                    K.bound_scales([dict(w=sc["dqkv_stats"], w_index=2, out_scale=sc["dqkv_scale"])])
                dw_later(dqkv, h1, gw, gb, arith=ar, dy_scale=None if o_h1 else dq_uni, x_scale=None if o_h1 else sc["h1_scale"])
                dh1 = K.linear_bwd_input(dqkv, wqkv, defer_reduce=fuse and i > 0,
                                         **prod(i, 4, a_scale=s_dqkv, b_scale=sc["cs_qkv"]))
            else:
                dw_later(dqkv, h1, gw, gb, arith=ar)
                dh1 = K.linear_bwd_input(dqkv, wqkv, arith=ar, defer_reduce=fuse and i > 0)
            g1w, g1b = G(b + "sublayer_connections.0.norm.weight"), G(b + "sublayer_connections.0.norm.bias")
            if fuse and i > 0:      # the gradient enters layer i - 1 through ITS FFN-output dropout: made here, with its scales
                s_dy2, bs_dz1 = (i32(), i32()) if scales is not None else (None, None)
                below = scales[i - 1] if scales is not None else None
                dx, dy2 = K.layernorm_bwd_dropout(dh1, x, W(b + "sublayer_connections.0.norm.weight"), mean1, rstd1, g1w, g1b,
                                                  dx2, p, seed, (i - 1) * 8 + _SITE_FFN_OUT, row_scale=s_dy2,
                                                  bound_factor=below["dz1_factor"] if below else None, bound_scale=bs_dz1,
                                                  row_scale_min=below["dy2_min"] if below else None,
                                                  bound_scale_min=below["dz1_min"] if below else None, pending=ln_pending,
                                                  planes=dy2_planes if below else None)
                have_min = below is not None
                have_planes = below is not None and dy2_planes is not None
            else:
                dx = K.layernorm_bwd(dh1, x, W(b + "sublayer_connections.0.norm.weight"), mean1, rstd1, g1w, g1b, dres=dx2,
                                     pending=ln_pending)
                dy2 = s_dy2 = bs_dz1 = None
                have_min = have_planes = False
            dw_flush()
            done(b + "self_attn.wq.weight", b + "sublayer_connections.1.norm.bias")
            ctx.saved[i] = None
        K.layernorm_bwd_flush(ln_pending)
        join()
        if ctx.measure:
            ctx.guard.submit(scales[0]["guard_stats"], scales[0]["ints"], scales[0]["minbuf"], scales)
        # ---- front end
        if not m.use_embedding:
            dx = K.posenc_add_bwd(dx, p, seed)
        convs = m.conv_shapes or []
        for j in reversed(range(len(convs))):
            ci, co, k = convs[j]
            xin, w2 = ctx.conv_saved[j]
            need_dx = m.use_embedding or j > 0                     # the one-hot input is data, not a parameter
            dx = K.conv1d_bwd(dx, xin, B, L, ci, w2, k, G(f"encoder.conv_layers.{j}.weight"),
                              G(f"encoder.conv_layers.{j}.bias"), need_dx=need_dx, arith=ar)
            ctx.conv_saved[j] = None
        if convs:
            done("encoder.conv_layers.0.weight", f"encoder.conv_layers.{len(convs) - 1}.bias")
        if m.use_embedding:
            K.embed_bwd(seq, dx, m.dmodel, p, seed, G("encoder.input_embedding.emb.weight"))
            done("encoder.input_embedding.emb.weight", "encoder.input_embedding.emb.weight")
        ctx.saved = ctx.conv_saved = ctx.x_last = ctx.pred = ctx.scales = None      # (whatever still refers to this node holds no activations)
        return None, None, None, None


# tokens x d_model below which AUTO runs the whole step in bf16x3: the scale bookkeeping of f16x2 costs more than its products
# save on a launch-bound step (config 1: 1.04 against 1.39 ms).  From 2^20 on f16x2 wins: 4.92 against 5.34 ms at 2048 x 512
# (with the LDS-DMA forward products), and since the producers of the staging GEMM were trimmed (end of round 3) also at
# 4096 x 256: 2.23 against 2.28 ms (config 2; it was 3.2 against 2.9 in round 2).
AUTO_F16X2_MIN_WORK = 1 << 20
# tokens x d_model below which the weight-gradient products of the backward pass run on a SIDE stream next to the dX chain
# (few output tiles per product: the persistent kernels leave CUs idle that the other stream's kernel can take).  Only for
# d_model >= 512 and >= 4096 tokens: measured -5 % at 8, -3 % at 16, -1.2 % at 32 proteins x 512, +-3 % at 4
# (profiles/r03/r03_ab_side_stream.txt), but +14 % on config 2 (4096 x 256) and +12 % on config 1, where the step is bound by
# host-side launches and the stream joins add to them
SIDE_STREAM_MAX_WORK = 1 << 23
SIDE_STREAM_MIN_TOKENS = int(os.environ.get("PTAMD_SIDE_MIN_TOKENS", 4096))     # (the environment variable: for measurements)
# tokens from which the products behind a LayerNorm run on ptamd_gemm_hp (below: ptamd_gemm, 128-row tiles)
HP_MIN_TOKENS = int(os.environ.get("PTAMD_HP_MIN_TOKENS", 4096))     # (the environment variable: for measurements)


class EncoderOnlyTransformer(_TransformerBase):
    """ A Transformer that only uses Encoder layers (reference: models/encoder_only.py:10). """

    def __init__(self, nlayers, nhead, dmodel, dff, max_seq_len, vocab, angle_means, use_tanh_out, dropout=0.1):
        super().__init__(nlayers, nhead, dmodel, dff, max_seq_len, vocab, angle_means, use_tanh_out, dropout=dropout)
