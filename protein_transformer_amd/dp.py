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
  * every reported loss is a statistic of the GLOBAL batch: the MSE term (`-l mse` / `combined`) is a mean over all
    selected elements of the global batch, so its numerator and count are SUM-reduced before the gradient is formed, and
    the dRMSD / RMSD statistics are reduced with them (losses.LossReport) - all ranks log, schedule and stop identically.
  * ragged batches are dealt to the ranks in serpentine order of length (`shard_indices`), each rank collates, pads (to
    its own longest protein) and uploads only its shard; a rank whose shard is empty still joins every collective.
"""
import os

import torch
import torch.distributed as dist


def is_initialized():
    return dist.is_available() and dist.is_initialized()


_SINGLE = 0          # > 0 inside `single_process()`: this process acts as a world of one (no collectives)


def world_size():
    return dist.get_world_size() if is_initialized() and not _SINGLE else 1


def rank():
    return dist.get_rank() if is_initialized() and not _SINGLE else 0


class single_process:
    """Context: inside it THIS process runs as a world of one - `world_size()` is 1, no collective is issued, a model's
    gradient hook is detached - while the process group stays up.  For self-checks that compare the data-parallel result
    with the same work done by one rank (`bench.py --verify-dp`); the other ranks must not enter a collective meanwhile."""

    def __init__(self, model=None):
        self.model, self.hook = model, None

    def __enter__(self):
        global _SINGLE
        _SINGLE += 1
        if self.model is not None:
            self.hook, self.model.grad_hook = self.model.grad_hook, None
        return self

    def __exit__(self, *exc):
        global _SINGLE
        _SINGLE -= 1
        if self.model is not None:
            self.model.grad_hook = self.hook
        return False


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
    """This rank's contiguous slice (along dim 0) of every tensor of a global batch (equal-length batches: bench, tests)."""
    w = world_size()
    if w == 1:
        return tensors
    lo, hi = shard_bounds(tensors[0].shape[0], w, rank())
    return tuple(t[lo:hi] for t in tensors)


def shard_indices(lengths, world=None, r=None):
    """Length-balanced deal of a ragged batch: positions of the proteins of `lengths` that rank
    `r` takes.  Proteins are sorted by length (longest first, ties by position) and dealt in serpentine order
    0..W-1, W-1..0, ... so that every rank gets the same number of proteins (+-1) and about the same sum of lengths and of
    squared lengths (the dRMSD cost).  Deterministic, identical on every rank, a partition of range(len(lengths));
    the positions of a rank come back in ascending order."""
    world = world_size() if world is None else world
    r = rank() if r is None else r
    if world == 1:
        return list(range(len(lengths)))
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    mine = []
    for pos, i in enumerate(order):
        rnd, k = divmod(pos, world)
        owner = k if rnd % 2 == 0 else world - 1 - k
        if owner == r:
            mine.append(i)
    return sorted(mine)


def broadcast_parameters(model, src=0):
    """Every rank takes rank `src`'s flat parameter buffer: data-parallel replicas must START identical - the SUM-reduced
    gradient keeps them identical from then on (bench.py's `param_checksum_spread` checks exactly that).  Anything that
    makes the ranks' initialisations differ (a seed, statistics of a rank's own data such as the angle means that
    initialise the output bias, encoder_only.py:28-33) is overwritten here."""
    if world_size() > 1:
        flat, _ = model.flat_parameters()
        dist.broadcast(flat, src=src)
        if hasattr(model, "weights_changed"):
            model.weights_changed()              # (a collective wrote the flat buffer: nothing prepared for the old weights holds)


def attach(model):
    """Overlap the gradient reduction with backward: reduce each slice of the flat gradient buffer as soon
    as `_EncoderFn.backward` reports it final (model.grad_hook).  The work handles live on the model.  The replicas are
    synchronised first (`broadcast_parameters`)."""
    model._dp_pending = []
    if world_size() == 1:
        model.grad_hook = None
        return model
    if hasattr(model, "flat_parameters") and getattr(model, "_flat_numel", None):
        broadcast_parameters(model)

    def hook(offset, numel):
        g = model._flat_grad
        if HOOK_TRACE is not None:         # (bench.py: which slice left from which stream, so a first RCCL run diagnoses itself)
            HOOK_TRACE.append(_hook_record(offset, numel, g.device))
        model._dp_pending.append(dist.all_reduce(g[offset:offset + numel], op=dist.ReduceOp.SUM, async_op=True))

    model.grad_hook = hook
    reserve_cus_for_collectives()
    return model


# Hook trace (bench.py `communication.hooks`): when HOOK_TRACE is a list every gradient hook appends what it reduced and the
# stream it was issued from - the collective orders itself behind THAT stream, so a slice that left
# from the wrong one is the first thing to look for when `param_checksum_spread` / `--verify-dp` are not zero.
HOOK_TRACE = None


def _hook_record(offset, numel, device):
    rec = {"offset": int(offset), "bytes": 4 * int(numel)}
    if device.type == "cuda":
        cur = torch.cuda.current_stream(device)
        rec["stream"] = "default" if cur == torch.cuda.default_stream(device) else hex(cur.cuda_stream)
    else:
        rec["stream"] = "host"
    return rec


def hook_trace_start():
    global HOOK_TRACE
    HOOK_TRACE = []


def hook_trace_stop():
    global HOOK_TRACE
    rec, HOOK_TRACE = HOOK_TRACE or [], None
    return rec


def reserve_cus_for_collectives(n=None):
    """The split-arithmetic GEMMs are persistent kernels that take every CU (one 512-thread workgroup with ~150 KB of LDS
    per CU): an RCCL kernel enqueued meanwhile shares the chip only with the non-GEMM kernels of the backward pass
    (attention, LayerNorm, loss) and the gaps between launches.  `n` > 0 (or PTAMD_DP_RESERVE_CUS) makes every GEMM leave
    n CUs free for it.  The default is 0: the benchmark's tile counts are multiples of 256 (T = 16384 tokens), so a grid
    of 256 - n workgroups needs a second round for the left-over tiles - a K = 512, N = 512 product would take twice as
    long; reserve only together with shapes that do not quantise on the full chip."""
    from . import kernels
    if n is None:
        n = int(os.environ.get("PTAMD_DP_RESERVE_CUS", "0"))
    kernels.GEMM_RESERVED_CUS = max(0, int(n))


# Exposed-communication meter (bench.py --gpus N): when COMM_METER is a list, every `all_reduce_gradients` brackets its
# waits with two events on the COMPUTE stream - the interval is the time the compute stream stood still for the
# reduction, i.e. what the overlap with backward did NOT hide - and appends (event, event, bytes reduced this step).
COMM_METER = None


def comm_meter_start():
    global COMM_METER
    COMM_METER = []


def comm_meter_stop():
    """-> (mean exposed wait per step in ms, bytes reduced per step) of the steps since `comm_meter_start`."""
    global COMM_METER
    rec, COMM_METER = COMM_METER or [], None
    if not rec:
        return 0.0, 0
    torch.cuda.synchronize()
    return sum(e0.elapsed_time(e1) for e0, e1, _ in rec) / len(rec), int(sum(b for _, _, b in rec) / len(rec))


def all_reduce_gradients(model, empty=False):
    """Finish the step's gradient exchange.  With `attach(model)` this only waits for the slices already
    in flight; without it the whole flat buffer is reduced here in one call.  `empty`: this rank had no proteins in
    the step (its gradient buffer is zero and no backward ran): it issues the same reductions, in the same order."""
    if world_size() == 1:
        return
    model.__dict__["_grad_dirty"] = True     # (a rank with an empty shard ran no backward pass, yet its buffer receives the sum)
    meter = COMM_METER is not None and torch.cuda.is_available()
    if meter:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if getattr(model, "grad_hook", None) is None:
        _, g = model.flat_parameters()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        nbytes = 4 * g.numel()
    else:
        if empty:
            for off, n in model.grad_slices():
                model.grad_hook(off, n)
        for work in model._dp_pending:
            work.wait()
        model._dp_pending.clear()
        nbytes = 4 * sum(n for _, n in model.grad_slices())
    if meter:
        e1.record()
        COMM_METER.append((e0, e1, nbytes))


def describe():
    """Who is in the job, as the job itself sees it (the self-check block of the bench line): backend, collective library
    version, and for every rank its device identity - gathered over the process group, so a rank that is missing, doubled
    or sitting on the same GPU as another one shows up here and not only in the scaling curve."""
    me = {"rank": rank(), "local_rank": local_rank(), "host": os.uname().nodename, "pid": os.getpid()}
    if torch.cuda.is_available():
        i = torch.cuda.current_device()
        pr = torch.cuda.get_device_properties(i)
        me.update(device=i, name=pr.name, uuid=str(getattr(pr, "uuid", "")), cus=int(pr.multi_processor_count),
                  pci=str(getattr(pr, "pci_bus_id", "")) + ":" + str(getattr(pr, "pci_device_id", "")))
    out = {"backend": dist.get_backend() if is_initialized() else "none (single process)", "world_size": world_size()}
    if out["backend"] == "nccl":
        try:
            out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:                                    # noqa: BLE001 - a version string is not worth a failed run
            out["rccl_version"] = f"unknown ({type(e).__name__})"
    if world_size() > 1:
        seen = [None] * world_size()
        dist.all_gather_object(seen, me)
    else:
        seen = [me]
    out["ranks_seen"] = seen
    ids = [(r.get("host"), r.get("uuid") or r.get("pci") or r.get("device")) for r in seen]
    out["distinct_devices"] = len(set(ids))
    out["ranks_ok"] = sorted(r["rank"] for r in seen) == list(range(world_size()))
    return out


def param_checksums(model):
    """(sum |p| in fp64, wrap-around int64 sum of the parameters' BIT PATTERNS) of the flat parameter buffer: two numbers
    that are equal on every rank iff the ranks hold the same parameters (the second one bit for bit, order-independent)."""
    flat, _ = model.flat_parameters()
    return float(flat.double().abs().sum().item()), int(flat.view(torch.int32).to(torch.int64).sum().item())


def param_checksum_spread(model):
    """max - min over the ranks of the two `param_checksums` (must both be 0: after the SUM all-reduce every rank applies
    the same update to the same parameters) + rank 0's values."""
    a, h = param_checksums(model)
    if world_size() == 1:
        return {"abs_sum_spread": 0.0, "bit_hash_spread": 0, "abs_sum": a, "bit_hash": h, "ranks": 1}
    every = [None] * world_size()
    dist.all_gather_object(every, (a, h))
    return {"abs_sum_spread": max(x[0] for x in every) - min(x[0] for x in every),
            "bit_hash_spread": max(x[1] for x in every) - min(x[1] for x in every),
            "abs_sum": every[0][0], "bit_hash": every[0][1], "ranks": len(every)}


def all_reduce_sum_(t):
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def barrier():
    if world_size() > 1:
        dist.barrier()
