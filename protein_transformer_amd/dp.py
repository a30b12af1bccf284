"""Data parallelism: one process per GPU, batch sharded by proteins, gradients SUM-all-reduced over RCCL.

The reference has no multi-GPU code at all ("Parallelism strategies"); its only
parallelism is a CPU worker pool for the loss (train.py:360-365, losses.py:144-147).  The MI355X
equivalent is new: proteins are independent units, so each rank takes a
contiguous shard of every batch and the only exchange step is ONE reduction of the flat fp32
gradient buffer per step.

  * SUM, not mean: the reference back-propagates the sum over proteins (losses.py:166-167), so the
    DP result equals the single-GPU result on the whole batch.
  * xGMI is point-to-point (7 links per GPU): the flat buffer is reduced in per-encoder-layer slices
    (12.6 MB each at d512) as soon as the backward pass has finished a layer, on RCCL's own stream,
    overlapped with the remaining backward kernels; `all_reduce_gradients` only waits for the tail.
  * gradient clipping needs the GLOBAL norm, so it runs after the reduction on every rank (identical).
"""
import os

import torch
import torch.distributed as dist

_pending = []


def is_initialized():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_initialized() else 1


def rank():
    return dist.get_rank() if is_initialized() else 0


def local_rank():
    """Device index of this process: LOCAL_RANK, folded onto the visible devices (so that a multi-rank smoke test
    can oversubscribe a single GPU; on an 8-GPU node it is the identity)."""
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    return lr % n if n else lr


def is_main():
    return rank() == 0


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun contract)."""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 or is_initialized():
        return
    backend = backend or os.environ.get("PTAMD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    # "nccl" is RCCL on ROCm; gloo is only for tests (CPU, or several ranks sharing one GPU)
    kwargs = {}
    if backend == "nccl":
        torch.cuda.set_device(local_rank())
        kwargs["device_id"] = torch.device("cuda", local_rank())
    dist.init_process_group(backend=backend, **kwargs)


def shutdown():
    if is_initialized():
        dist.destroy_process_group()


def shard_bounds(n, world, r):
    """Contiguous split of n proteins over `world` ranks; the first n % world ranks take one extra."""
    base, extra = divmod(n, world)
    lo = r * base + min(r, extra)
    return lo, lo + base + (1 if r < extra else 0)


def shard_batch(*tensors):
    """This rank's contiguous slice (along dim 0) of every tensor of a global batch."""
    w = world_size()
    if w == 1:
        return tensors
    lo, hi = shard_bounds(tensors[0].shape[0], w, rank())
    return tuple(t[lo:hi] for t in tensors)


def attach(model):
    """Overlap the gradient reduction with backward: reduce each slice of the flat gradient buffer as soon
    as `_EncoderFn.backward` reports it final (model.grad_hook)."""
    if world_size() == 1:
        model.grad_hook = None
        return model

    def hook(offset, numel):
        _, g = model._flat, model._flat_grad
        _pending.append(dist.all_reduce(g[offset:offset + numel], op=dist.ReduceOp.SUM, async_op=True))

    model.grad_hook = hook
    return model


def all_reduce_gradients(model):
    """Finish the step's gradient exchange.  With `attach(model)` this only waits for the slices already
    in flight; without it the whole flat buffer is reduced here in one call."""
    if world_size() == 1:
        return
    if model.grad_hook is None:
        _, g = model.flat_parameters()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return
    for work in _pending:
        work.wait()
    _pending.clear()


def all_reduce_sum_(t):
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def barrier():
    if world_size() > 1:
        dist.barrier()
