"""Shared pieces of the GPU parity tests (TEST INFRASTRUCTURE: imports the oracle).

  * `fp64_reference(model, seq, crd, nhead)`: predictions, per-protein losses and the parameter gradient of one step in an
    fp64 evaluation of the oracle's formulas on the model's current weights (oracle.encoder + oracle.batched);
  * `Fp64Trainer`: the same step repeated - fp64 weights, `torch.optim.SGD / Adam` and `clip_grad_norm_` in fp64 with the
    reference's hyper-parameters (train.py:41-46,371-381) - the trajectory a device run is compared with;
  * `grad_errors(model, ref)`: relative L2 error of the device gradient, whole vector and per parameter group.
"""
import numpy as np
import torch

_POOL = None


def cpu_pool():
    """A spawn pool for the CPU side of the parity tests (round 6: the fp64 NeRF builds of the parity record - Python loops
    over chains of up to 1500 residues - ran one after the other and were 260 s of the GPU suite); one per test process."""
    global _POOL
    if _POOL is None:
        import atexit
        import multiprocessing as mp
        import os
        n = max(1, min(16, (os.cpu_count() or 2) // 2))
        _POOL = mp.get_context("spawn").Pool(n, initializer=torch.set_num_threads, initargs=(1,))
        atexit.register(_POOL.terminate)
    return _POOL


def _build_coords_job(job):
    from oracle import batched as obat
    rad, seq, dtype_name = job
    dt = getattr(torch, dtype_name)
    return obat.generate_coords_batched(torch.from_numpy(rad).to(dt), torch.from_numpy(seq), dt).double().numpy()


def build_coords_many(jobs):
    """jobs: list of (angles [B, L, 12] tensor, sequences [B, L] tensor, torch dtype) -> list of fp64 numpy coordinates, built by
    oracle.batched.generate_coords_batched in the pool (in order)."""
    packed = [(r.detach().cpu().double().numpy(), q.detach().cpu().numpy(), str(dt).split(".")[-1]) for r, q, dt in jobs]
    if len(packed) == 1:
        return [_build_coords_job(packed[0])]
    return cpu_pool().map(_build_coords_job, packed)


def group_of(name):
    if "input_embedding" in name:
        return "embedding"
    if "conv_layers" in name:
        return "conv." + ("weight" if name.endswith("weight") else "bias")
    if "output_projection" in name:
        return "out." + ("weight" if name.endswith("weight") else "bias")
    if "norm" in name:
        return "layernorm." + ("gain" if name.endswith("weight") else "bias")
    if "self_attn" in name:
        return "attention." + ("weight" if name.endswith("weight") else "bias")
    return "ffn." + ("weight" if name.endswith("weight") else "bias")


def fp64_reference(params, seq, crd, nhead):
    """params: name -> fp64 CPU tensor (reference keys).  Returns dict(pred, rad, stats, crd, grads)."""
    from oracle import batched as obat
    from oracle import encoder as oenc
    B, L = seq.shape
    pe = {k: v for k, v in params.items() if k.endswith(".pe")}
    leaf = {k: v.detach().clone().requires_grad_() for k, v in params.items() if k not in pe}
    pred = oenc.encoder_forward({**leaf, **pe}, seq, nhead)
    cs = pred.view(B, L, 12, 2)
    rad = torch.atan2(cs[..., 1], cs[..., 0])
    stats, crd64, dang = obat.batch_loss_and_grads(rad, seq, crd, dtype=torch.float64)
    rad.backward(dang)
    return {"pred": pred.detach(), "rad": rad.detach(), "stats": stats, "crd": crd64,
            "grads": {k: v.grad for k, v in leaf.items()}, "radius": torch.sqrt(cs[..., 1] ** 2 + cs[..., 0] ** 2).detach()}


class Fp64Trainer:
    """One process, fp64: `step(seq, crd)` = zero_grad, forward, sum_i lndrmsd_i backward, clip, optimizer step."""

    def __init__(self, params, nhead, optimizer="sgd", lr=1e-4, clip=1.0, weight_decay=10e-3):
        self.pe = {k: v.double() for k, v in params.items() if k.endswith(".pe")}
        self.params = {k: v.detach().double().clone().requires_grad_() for k, v in params.items() if not k.endswith(".pe")}
        self.nhead, self.clip = nhead, clip
        ps = list(self.params.values())
        self.opt = (torch.optim.Adam(ps, betas=(0.9, 0.98), eps=1e-9, lr=lr, weight_decay=weight_decay) if optimizer == "adam"
                    else torch.optim.SGD(ps, lr=lr, weight_decay=weight_decay))

    def step(self, seq, crd):
        from oracle import batched as obat
        from oracle import encoder as oenc
        B, L = seq.shape
        self.opt.zero_grad()
        pred = oenc.encoder_forward({**self.params, **self.pe}, seq, self.nhead)
        cs = pred.view(B, L, 12, 2)
        rad = torch.atan2(cs[..., 1], cs[..., 0])
        stats, _, dang = obat.batch_loss_and_grads(rad, seq, crd, dtype=torch.float64)
        rad.backward(dang)
        if self.clip:
            torch.nn.utils.clip_grad_norm_(list(self.params.values()), self.clip)
        self.opt.step()
        return {"drmsd": float(np.mean([s[0] for s in stats])), "lndrmsd": float(np.mean([s[1] for s in stats]))}

    def state(self):
        return {**{k: v.detach() for k, v in self.params.items()}, **self.pe}


def grad_errors(named_grads, ref):
    """named_grads / ref: name -> tensor.  -> (rel-L2 of the whole vector, {group: rel-L2}, (worst tensor, its rel-L2))."""
    got = {n: g.detach().cpu().double() for n, g in named_grads.items()}
    num = sum(float(((got[n] - ref[n]) ** 2).sum()) for n in ref)
    den = sum(float((ref[n] ** 2).sum()) for n in ref)
    groups, worst = {}, ("", 0.0)
    for n in ref:
        g = groups.setdefault(group_of(n), [0.0, 0.0])
        e2, r2 = float(((got[n] - ref[n]) ** 2).sum()), float((ref[n] ** 2).sum())
        g[0] += e2
        g[1] += r2
        if r2 > 1e-24 * den and (e2 / r2) ** 0.5 > worst[1]:
            worst = (n, (e2 / r2) ** 0.5)
    return (num / den) ** 0.5, {k: (v[0] / v[1]) ** 0.5 for k, v in groups.items() if v[1] > 0}, worst


def params_rel_l2(state_dev, state_ref):
    """|| theta_dev - theta_ref || / || theta_ref || over every parameter (reference keys, positional table excluded)."""
    num = den = 0.0
    for k, v in state_ref.items():
        if k.endswith(".pe"):
            continue
        d = state_dev[k].detach().cpu().double() - v.double()
        num += float((d ** 2).sum())
        den += float((v.double() ** 2).sum())
    return (num / den) ** 0.5


def update_record(path, key, value):
    import json
    import os
    os.makedirs(os.path.dirname(path), exist_ok=True)
    rec = {}
    if os.path.exists(path):
        with open(path) as f:
            rec = json.load(f)
    rec[key] = value
    with open(path, "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
