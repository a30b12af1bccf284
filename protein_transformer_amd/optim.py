"""Fused clip + optimizer step over the model's flat parameter buffer.

Replaces `torch.nn.utils.clip_grad_norm_` + `torch.optim.SGD/Adam.step()` as used by the reference
(/root/reference/protein_transformer/train.py:41-46,371-381) with three kernel launches over one
flat fp32 buffer (csrc/optim.hip) and no host synchronisation.  Both classes are
`torch.optim.Optimizer`s, so LR schedulers (ReduceLROnPlateau, the Noam wrapper) and
`state_dict()` work unchanged; hyper-parameters are read from `param_groups[0]` every step.
"""
import numpy as np
import torch

from . import kernels as K


class _FlatOptimizer(torch.optim.Optimizer):
    def __init__(self, model, defaults):
        self.model = model
        super().__init__(list(model.parameters()), defaults)
        self._sqnorm = None
        self.max_norm = 0.0
        # True: `step()` zeroes the gradient buffer behind its last read and the `zero_grad()` that opens the next step
        # (train.py:37) finds nothing to do - the loop of train.train_step / the reference's train_epoch, where nobody looks
        # at the gradients between `optimizer.step()` and the next `zero_grad()`.  False (the default of the classes: torch's
        # semantics, `p.grad` survives the step); train.setup_model_optimizer_scheduler switches it on.
        self.zero_grad_in_step = False
        self._zeroed = None                      # (address, version) of the gradient buffer as the last step left it: all zero

    def zero_grad(self, set_to_none=False):
        _, g = self.model.flat_parameters()
        # nothing to do when the last step zeroed this very buffer and nothing has written it since: no backward pass of the
        # model (raw-pointer writes: `_grad_dirty`), no torch op on it or on a `p.grad` view (their shared version counter)
        if (self._zeroed == (g.data_ptr(), g._version) and not self.model.__dict__.get("_grad_dirty", True)):
            return
        g.zero_()
        self._zeroed = None

    def _after_step(self, g, zeroed):
        if zeroed:
            self.model.__dict__["_grad_dirty"] = False
            self._zeroed = (g.data_ptr(), g._version)
        else:
            self._zeroed = None

    def clip_grad_norm_(self, max_norm):
        """Device-side `clip_grad_norm_`: launches the squared-norm reduction and arms the next `step()`
        to scale gradients by min(1, max_norm / (norm + 1e-6)).  Returns the 1-element device tensor
        holding ||g||^2 (no sync; take .sqrt().item() if the value is wanted on the host)."""
        _, g = self.model.flat_parameters()
        if self._sqnorm is None or self._sqnorm.device != g.device:
            self._sqnorm = torch.zeros(1, dtype=torch.float32, device=g.device)
        K.grad_sqnorm(g, self._sqnorm)
        self.max_norm = float(max_norm or 0.0)
        return self._sqnorm

    def _prepared(self):
        """(plan, with_planes, mark_fresh) when the model wants the step to prepare the next forward pass's view of the weights
        (models/encoder_only.py `prepared_step`), else None: the plain optimizer kernel."""
        fn = getattr(self.model, "prepared_step", None)
        return fn() if fn is not None else None

    def _written(self):
        """The step wrote the flat buffer through a raw pointer: every cached view of the weights (scales, bounds, planes - of
        the training pass AND of the evaluation pass) is stale from here on (models/encoder_only.py `weights_written`)."""
        fn = getattr(self.model, "weights_written", None)
        if fn is not None:
            fn()

    def _clip_args(self):
        sq, mx = (self._sqnorm, self.max_norm) if self.max_norm > 0 else (None, 0.0)
        self.max_norm = 0.0                      # one clip arms one step, like the reference's call order
        return sq, mx


class FusedSGD(_FlatOptimizer):
    """SGD with L2 weight decay (torch.optim.SGD semantics, train.py:379-381)."""

    def __init__(self, model, lr=1e-4, weight_decay=0.0):
        super().__init__(model, dict(lr=lr, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        w, g = self.model.flat_parameters()
        grp = self.param_groups[0]
        sq, mx = self._clip_args()
        prepared = self._prepared()
        zero = bool(self.zero_grad_in_step)
        if prepared is None:
            K.sgd_step(w, g, sq, mx, grp["lr"], grp["weight_decay"], zero_grad=zero)
            self._written()
        else:       # the same update, and the scales / bounds / planes of the NEW weights for the next forward pass (csrc/wprep.hip)
            plan, with_planes, mark_fresh = prepared
            plan.sgd_step(w, g, sq, mx, grp["lr"], grp["weight_decay"], with_planes=with_planes, zero_grad=zero)
            self._written()
            mark_fresh()        # (after the bump: only the cache this step prepared carries the new stamp)
        self._after_step(g, zero)


class FusedAdam(_FlatOptimizer):
    """Adam with L2 weight decay (torch.optim.Adam semantics, train.py:374-378)."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.0):
        super().__init__(model, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._m = self._v = None
        self._t = 0

    @torch.no_grad()
    def step(self, closure=None):
        w, g = self.model.flat_parameters()
        if self._m is None:
            self._m, self._v = torch.zeros_like(w), torch.zeros_like(w)
        elif self._m.device != w.device:      # moments loaded from a checkpoint (CPU) or a model moved since: migrate
            self._m, self._v = self._m.to(w.device), self._v.to(w.device)
        grp = self.param_groups[0]
        self._t += 1
        sq, mx = self._clip_args()
        prepared = self._prepared()
        zero = bool(self.zero_grad_in_step)
        if prepared is None:
            K.adam_step(w, g, self._m, self._v, sq, mx, grp["lr"], grp["betas"][0], grp["betas"][1], grp["eps"],
                        grp["weight_decay"], self._t, zero_grad=zero)
            self._written()
        else:
            # (ptamd_adam_step_prep - the fused form, tested in tests/test_gpu_scales.py - streams seven arrays through a kernel
            # that holds whole rows in registers: 190 us against 84 + 61 for the plain Adam kernel followed by the preparation
            # pass over the new weights, profiles/r05/r05_wprep_bench.txt; the SGD form breaks even and saves a launch)
            plan, with_planes, mark_fresh = prepared
            K.adam_step(w, g, self._m, self._v, sq, mx, grp["lr"], grp["betas"][0], grp["betas"][1], grp["eps"],
                        grp["weight_decay"], self._t, zero_grad=zero)
            self._written()
            plan.prepare(w, with_planes=with_planes)
            mark_fresh()
        self._after_step(g, zero)

    def state_dict(self):
        sd = super().state_dict()
        sd["flat_state"] = dict(step=self._t, exp_avg=self._m, exp_avg_sq=self._v)
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        fs = sd.pop("flat_state", None)
        super().load_state_dict(sd)
        if fs is not None:
            self._t, self._m, self._v = int(fs["step"]), fs["exp_avg"], fs["exp_avg_sq"]
            if self._m is not None:           # keep the moments: move them next to the flat parameter buffer
                params = list(self.model.parameters())
                dev = params[0].device if params else self._m.device
                self._m = self._m.detach().to(dev, torch.float32).contiguous().clone()
                self._v = self._v.detach().to(dev, torch.float32).contiguous().clone()


class ScheduledOptim():
    """Noam learning-rate schedule wrapper, API of the reference's
    models/transformer/Optimizer.py:4-62: lr = d_model^-0.5 * min(step^-0.5, step * warmup^-1.5)."""

    def __init__(self, optimizer, d_model, n_warmup_steps):
        self._optimizer = optimizer
        self.n_warmup_steps = n_warmup_steps
        self.n_current_steps = 0
        self.init_lr = np.power(d_model, -0.5)

    def step(self):
        self._update_learning_rate()
        self._optimizer.step()

    def zero_grad(self):
        self._optimizer.zero_grad()

    def clip_grad_norm_(self, max_norm):
        return self._optimizer.clip_grad_norm_(max_norm)

    def _get_lr_scale(self):
        return np.min([np.power(self.n_current_steps, -0.5),
                       np.power(self.n_warmup_steps, -1.5) * self.n_current_steps])

    def _update_learning_rate(self):
        self.n_current_steps += 1
        lr = self.init_lr * self._get_lr_scale()
        self.cur_lr = lr
        for param_group in self._optimizer.param_groups:
            param_group['lr'] = lr

    @property
    def param_groups(self):
        return self._optimizer.param_groups

    @property
    def zero_grad_in_step(self):
        return self._optimizer.zero_grad_in_step

    @zero_grad_in_step.setter
    def zero_grad_in_step(self, value):
        self._optimizer.zero_grad_in_step = bool(value)

    def state_dict(self):
        return (self._optimizer.state_dict(), self.n_warmup_steps, self.n_current_steps, self.init_lr)

    def load_state_dict(self, d):
        self._optimizer.load_state_dict(d[0])
        self.n_warmup_steps, self.n_current_steps, self.init_lr = d[1], d[2], d[3]
