"""Training driver for the MI355X hot path; command-line compatible with the reference's train.py.

Mirrors /root/reference/protein_transformer/train.py: `train_epoch` :28-54, `get_losses` :57-111,
`eval_epoch` :114-135, `train` :138-186, `checkpoint_model` :189-230, `load_model` :233-271,
`make_model` :274-321, `seed_rngs` :340-357, `init_worker_pool` :360-365,
`setup_model_optimizer_scheduler` :368-393, `create_parser` :396-529, `main` :553-676.
Same flags plus a few additions at the bottom of the parser: max sequence
length, synthetic data, wandb on/off.  One process drives one GPU; under
`python -m torch.distributed.run` every rank takes a shard of each batch and gradients are
SUM-all-reduced over RCCL (protein_transformer_amd/dp.py).

What is different by design: the whole step stays on the device (no CPU loss workers, so
`init_worker_pool` returns None and `--sequential_drmsd_loss` is a no-op), clip + optimizer are one
fused kernel sequence, wandb / PyMOL are optional.
"""
import argparse
import csv
import os
import random
import sys
import time

import numpy as np
import torch

from . import dp
from .dataset import MAX_SEQ_LEN, DevicePrefetcher, prepare_dataloaders
from .log import (EarlyStoppingCondition, do_eval_batch_logging, do_eval_epoch_logging, do_train_batch_logging,
                  init_metrics, log_batch, prepare_log_header, reset_metrics_for_epoch, update_loss_trackers,
                  update_metrics_end_of_epoch)
from .losses import LossReport, batch_loss, combine_drmsd_mse, mse_grad, mse_sums
from .models.convolutional_encoder import ConvEncoderOnlyTransformer
from .models.encoder_only import EncoderOnlyTransformer
from .optim import FusedAdam, FusedSGD, ScheduledOptim
from .protein.Sequence import VOCAB
from .protein.Structure import NUM_PREDICTED_ANGLES, raise_for_status

START_EPOCH = 0
START_TIME = time.time()


def train_epoch(model, training_data, validation_datasets, optimizer, device, args, log_writer, metrics, pool=None):
    """ One complete training epoch (train.py:28-54).  Under data parallelism the loader already yields this rank's
    shard of every batch (dataset.ShardedBatchSampler): only that shard is collated and uploaded. """
    model.train()
    metrics = reset_metrics_for_epoch(metrics, "train")
    # batches arrive one ahead: the next upload runs on a side stream under this step (dataset.DevicePrefetcher); the
    # residue count is taken on the host before the upload - no device synchronisation in the loop
    for step, (src_seq, tgt_ang, tgt_crds, n_res) in enumerate(DevicePrefetcher(training_data, device)):
        losses = train_step(model, optimizer, args, src_seq, tgt_ang, tgt_crds, pool=pool, n_res=n_res)
        metrics = do_train_batch_logging(metrics, losses, src_seq, optimizer, args, log_writer, START_TIME, step)
        if (getattr(args, "structure_dir", None) and dp.is_main() and src_seq.shape[0]
                and step % max(1, args.log_structure_step) == 0):
            dump_structure(model, args, src_seq, tgt_crds, step)
    metrics = update_metrics_end_of_epoch(metrics, "train")
    return metrics


def dump_structure(model, args, src_seq, tgt_crds, step, struct_name="train"):
    """Structure dump of the first protein of a batch (log.py:158,201-206 of the reference): predict its angles again
    without dropout, build the atoms with the NeRF kernels and write the PDB files."""
    from .log import log_structure
    from .losses import inverse_trig_transform
    from .protein.Structure import generate_coords
    n = int((src_seq[0] != VOCAB.pad_id).sum().item())
    was_training = model.training
    model.eval()
    with torch.no_grad():
        pred = model(src_seq[:1], None)
        ang = inverse_trig_transform(pred)[0, :n]
        crd = generate_coords(ang, src_seq[0, :n], src_seq.device)
    model.train(was_training)
    return log_structure(args, crd, tgt_crds[0, :n * 14], src_seq[0, :n], step, struct_name)


def train_step(model, optimizer, args, src_seq, tgt_ang, tgt_crds, pool=None, n_res=None):
    """The body of the reference's training loop (train.py:36-46) for one device-resident batch:
    zero_grad, forward, losses + backward, (gradient all-reduce), clip, optimizer step.
    A rank whose shard of the batch is empty (fewer proteins than ranks) skips forward / backward and joins the
    collectives with zeros."""
    optimizer.zero_grad()
    empty = src_seq.shape[0] == 0
    pred = None if empty else model(src_seq, tgt_ang)
    losses = get_losses(args, pred, tgt_ang, tgt_crds, src_seq, pool=pool, n_res=n_res)
    dp.all_reduce_gradients(model, empty=empty)
    if args.clip:
        optimizer.clip_grad_norm_(args.clip)
    optimizer.step()
    return losses


def get_losses(args, pred, tgt_ang, tgt_crds, src_seq, pool=None, log=True, do_backwards=True, return_rmsd=False,
               eval_mode=False, n_res=None):
    """Losses/metrics of a batch (train.py:57-111); `loss` depends on `args.loss`.  Returns the reference's dictionary
    of 10 entries as HOST numbers (np.float64 / 0-d CPU tensors), obtained with one asynchronous copy that is waited
    for only after the backward pass has been enqueued (losses.LossReport) - no other device synchronisation.

    Gradient bookkeeping is the reference's: the dRMSD term back-propagates the SUM
    over proteins of the length-normalised loss whatever the reported loss is; `combined` adds
    (1-w)/0.01 * d(mse); `mse` back-propagates the MSE only.  Where the reference runs two backward
    passes through the model for `combined` (retain_graph), the two gradients are added first and
    the model is traversed once.

    Data parallel: `pred`, the targets and `src_seq` are this rank's shard; every returned value is a statistic of
    the GLOBAL batch and the MSE gradient is normalised by the global count of selected elements, so that the SUM of
    the ranks' parameter gradients is the single-process gradient.  `pred` may be None for an
    empty shard.
    """
    dev = src_seq.device
    empty = src_seq.shape[0] == 0
    need_drmsd = args.loss in ["lndrmsd", "drmsd", "combined"] or eval_mode
    if getattr(args, "backbone_loss", False) and need_drmsd:
        raise NotImplementedError("--backbone_loss is broken in the reference too (SURVEY.md A-4)")
    sums = stats = grad = status = rmsd = None
    if not empty:
        sums = mse_sums(pred, tgt_ang)                         # the three MSEs of train.py:64-66 in one pass
        if need_drmsd:
            stats, grad, status, crd = batch_loss(pred, tgt_crds, src_seq, do_backward=do_backwards, return_crd=True)
            if return_rmsd:
                from .eval_metrics import kabsch_rmsd_batch
                rmsd = kabsch_rmsd_batch(crd, tgt_crds, src_seq)
    report = LossReport(dev, stats=stats, status=status, mse_sums_local=sums, rmsd=rmsd, n_res=n_res)
    if do_backwards and not empty:
        w = args.combined_drmsd_weight
        if args.loss == "mse":
            g = mse_grad(pred, tgt_ang, report.global_mse_sums, coef=1.0)
        elif args.loss == "combined":
            g = mse_grad(pred, tgt_ang, report.global_mse_sums, coef=(1 - w) / 0.01, accumulate_into=grad.view_as(pred))
        else:
            g = grad
        pred.backward(gradient=g.view_as(pred))
    host = report.wait()
    if need_drmsd:
        raise_for_status(host["status"], theta_is_error=False)
    m = host["mse"] if host["mse"] is not None else np.full(6, np.nan)
    with np.errstate(invalid="ignore", divide="ignore"):
        m_loss_full, m_loss_bb, m_loss_sc = (torch.tensor(m[2 * k] / m[2 * k + 1], dtype=torch.float32) for k in range(3))
    rmsd_loss = host["rmsd"] if return_rmsd else None
    if need_drmsd:
        d_loss, ln_d_loss, d_bb_loss, d_bb_ln_loss = (np.float64(host[k]) for k in ("drmsd", "lndrmsd", "drmsd-bb", "lndrmsd-bb"))
        c_loss = combine_drmsd_mse(ln_d_loss, m_loss_full, w=args.combined_drmsd_weight, log=log)
        if args.loss == "lndrmsd":
            loss = ln_d_loss
        elif args.loss == "drmsd":
            loss = d_loss
        elif args.loss == "combined":
            loss = c_loss
        else:
            loss = m_loss_full
    else:
        d_loss, ln_d_loss, d_bb_loss, d_bb_ln_loss, c_loss = (torch.tensor(0),) * 5
        loss = m_loss_full
    out = {"loss": loss, "drmsd-full": d_loss, "lndrmsd-full": ln_d_loss, "drmsd-bb": d_bb_loss,
           "lndrmsd-bb": d_bb_ln_loss, "combined-full": c_loss, "mse-full": m_loss_full, "mse-bb": m_loss_bb,
           "mse-sc": m_loss_sc, "rmsd-full": rmsd_loss}
    if host["n_res"] is not None:
        out["n-residues"] = host["n_res"]                      # residues of the GLOBAL batch (speed meter, log.py)
    return out


def eval_epoch(model, validation_data, device, args, metrics, mode="valid", pool=None):
    """ One complete evaluation epoch (train.py:114-135); sharded over the ranks like training. """
    model.eval()
    metrics = reset_metrics_for_epoch(metrics, mode)
    with torch.no_grad():
        for src_seq, tgt_ang, tgt_crds, n_res in DevicePrefetcher(validation_data, device):
            pred = model(src_seq, tgt_ang) if src_seq.shape[0] else None
            losses = get_losses(args, pred, tgt_ang, tgt_crds, src_seq, pool=pool, do_backwards=False,
                                eval_mode=True, return_rmsd=True, n_res=n_res)
            metrics = do_eval_batch_logging(metrics, losses, src_seq, args, mode)
    do_eval_epoch_logging(metrics, mode)
    return metrics


def train(model, metrics, training_data, train_eval_loader, validation_datasets, test_data, optimizer, device, args,
          log_writer, scheduler, drmsd_worker_pool):
    """ Model training control loop (train.py:138-186). """
    for epoch_i in range(START_EPOCH, args.epochs):
        if dp.is_main():
            print(f'[ Epoch {epoch_i} ]')
        metrics["epoch"] = epoch_i
        metrics = train_epoch(model, training_data, validation_datasets, optimizer, device, args, log_writer, metrics,
                              pool=drmsd_worker_pool)
        if args.eval_train:
            metrics = eval_epoch(model, train_eval_loader, device, args, metrics, mode="train", pool=drmsd_worker_pool)
        if dp.is_main():
            log_batch(log_writer, metrics, START_TIME, mode="train", end_of_epoch=True)
        if not args.train_only:
            for split, validation_data in validation_datasets.items():
                metrics = eval_epoch(model, validation_data, device, args, metrics, mode=f"valid-{split}",
                                     pool=drmsd_worker_pool)
                if dp.is_main():
                    log_batch(log_writer, metrics, START_TIME, mode=f"valid-{split}", end_of_epoch=True)
        # every rank holds the same (globally reduced) metrics, so the scheduler, the early-stopping test and the
        # checkpoint policy below take the same branch on every rank
        if scheduler:
            scheduler.step(metrics[args.es_mode][f"epoch-{args.es_metric}-full"])
        try:
            metrics = update_loss_trackers(args, epoch_i, metrics)
        except EarlyStoppingCondition:
            break
        if dp.is_main():
            checkpoint_model(args, optimizer, model, metrics, epoch_i, scheduler)
    if not args.train_only and test_data is not None:
        metrics = eval_epoch(model, test_data, device, args, metrics, mode="test", pool=drmsd_worker_pool)
        if dp.is_main():
            log_batch(log_writer, metrics, START_TIME, mode="test", end_of_epoch=True)
    return metrics


def checkpoint_model(args, optimizer, model, metrics, epoch_i, scheduler):
    """Records model state according to the reference's checkpointing policy (train.py:189-230): `<chkpt_path>_best.chkpt`
    when this epoch's loss beats every earlier one, else `_latest.chkpt` when --checkpoint_time_interval hours have
    passed since the last checkpoint, else nothing.  Same dictionary layout.  Returns True iff the model was saved."""
    cur_loss, loss_history = metrics["loss_to_compare"], metrics["losses_to_compare"]
    if args.checkpoint_time_interval == 0:
        do_time_chkpt = False
    else:
        do_time_chkpt = (time.time() - metrics["last_chkpt_time"]) / 3600 > args.checkpoint_time_interval
    if len(loss_history) == 1 or cur_loss < min(loss_history[:-1]):
        modifier = "best"
    elif do_time_chkpt:
        modifier = "latest"
    else:
        return False
    chkpt_file_name = args.chkpt_path + f"_{modifier}.chkpt"
    checkpoint = {'model_state_dict': model.state_dict(), 'settings': args, 'epoch': epoch_i,
                  'optimizer_state_dict': optimizer.state_dict(),
                  'scheduler_state_dict': scheduler.state_dict() if scheduler else None,
                  'loss': cur_loss, 'metrics': metrics, 'elapsed_time': time.time() - START_TIME}
    torch.save(checkpoint, chkpt_file_name)
    metrics["last_chkpt_time"] = time.time()
    print('\r    - [Info] The checkpoint file has been updated.')
    return True


def load_model(model, optimizer, scheduler, args):
    """Resume from `<chkpt_path>_best.chkpt` (or --load_chkpt) unless --restart (train.py:233-271).  START_TIME is moved
    back by the checkpoint's elapsed time so that the `time` column of the log stays cumulative."""
    global START_EPOCH, START_TIME
    path = args.load_chkpt if getattr(args, "load_chkpt", None) else args.chkpt_path + "_best.chkpt"
    if args.restart or not os.path.exists(path):
        return model, optimizer, scheduler, False, init_metrics(args)
    if dp.is_main():
        print(f"[Info] Attempting to load model from {path}.")
    checkpoint = torch.load(path, map_location="cpu", weights_only=False)
    model.load_state_dict(checkpoint['model_state_dict'])
    if not args.restart_opt:
        optimizer.load_state_dict(checkpoint['optimizer_state_dict'])
    if scheduler and checkpoint.get('scheduler_state_dict'):
        scheduler.load_state_dict(checkpoint['scheduler_state_dict'])
    START_EPOCH = checkpoint['epoch'] + 1
    START_TIME -= checkpoint['elapsed_time']
    if dp.is_main():
        print(f"[Info] Resuming model training from end of Epoch {checkpoint['epoch']}. Previous validation loss"
              f" = {checkpoint['loss']:.4f}.")
    return model, optimizer, scheduler, True, checkpoint['metrics']


def make_model(args, device, angle_means):
    """Requested architecture (train.py:274-321): `enc-only` or `conv-enc[|k1,k2,k3|r1,r2,r3]`."""
    if args.model == "enc-only":
        return EncoderOnlyTransformer(nlayers=args.n_layers, nhead=args.n_head, dmodel=args.d_model,
                                      dff=args.d_inner_hid, max_seq_len=args.max_seq_len, dropout=args.dropout,
                                      vocab=VOCAB, angle_means=angle_means, use_tanh_out="linear-out" not in args.model)
    if "conv-enc" in args.model:
        sizes = [a for a in [getattr(args, "conv1_size", None), getattr(args, "conv2_size", None),
                             getattr(args, "conv3_size", None)] if a]
        reducs = [a for a in [getattr(args, "conv1_reduc", None), getattr(args, "conv2_reduc", None),
                              getattr(args, "conv3_reduc", None)] if a]
        return ConvEncoderOnlyTransformer(nlayers=args.n_layers, nhead=args.n_head, dmodel=args.d_model,
                                          dff=args.d_inner_hid, max_seq_len=args.max_seq_len, dropout=args.dropout,
                                          vocab=VOCAB, angle_means=angle_means,
                                          use_tanh_out="linear-out" not in args.model, conv_kernel_sizes=sizes,
                                          conv_dim_reductions=reducs, use_embedding=args.use_embedding,
                                          conv_out_matches_dm=args.conv_out_matches_dm)
    raise argparse.ArgumentError(None, "Model architecture not implemented on the MI355X path "
                                       "(enc-dec is deprecated upstream, README.md:49).")


def parse_conv_kernel_info_from_model_name(mname):
    """ "conv-enc|3,7,11|2,2,2" -> ([3, 7, 11], [2.0, 2.0, 2.0])   (train.py:323-338) """
    try:
        _, kernel_sizes, dim_reducs = mname.split("|")
    except ValueError:
        return [], []
    return list(map(int, kernel_sizes.split(","))), list(map(float, dim_reducs.split(",")))


def seed_rngs(args):
    """ Seed all necessary random number generators (train.py:340-357). """
    seed = args.seed
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def init_worker_pool(args):
    """The reference spawns cpu_count() loss workers (train.py:360-365); the loss runs on the GPU here."""
    return None


def setup_model_optimizer_scheduler(args, device, angle_means):
    """train.py:368-393: model, SGD/Adam with weight decay 0.01, Noam or plateau scheduling."""
    if not hasattr(args, "max_seq_len"):
        args.max_seq_len = MAX_SEQ_LEN
    model = make_model(args, device, angle_means).to(device)
    model.dropout_seed = (args.seed * 0x9E3779B1 + 7919 * dp.rank()) & (2 ** 62 - 1)
    wd = 10e-3 if args.weight_decay else 0
    if args.optimizer == "adam":
        optimizer = FusedAdam(model, betas=(0.9, 0.98), eps=1e-09, lr=args.learning_rate, weight_decay=wd)
    elif args.optimizer == "sgd":
        optimizer = FusedSGD(model, lr=args.learning_rate, weight_decay=wd)
    # this module's loop is zero_grad -> forward -> backward -> clip -> step (train_step = train.py:36-46) and nothing reads the
    # gradients behind the step: the step zeroes them itself and the next zero_grad finds nothing to do (optim.py)
    optimizer.zero_grad_in_step = True
    if args.lr_scheduling == "noam":
        optimizer = ScheduledOptim(optimizer, args.d_model, args.n_warmup_steps)
        scheduler = None
    else:
        scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, patience=args.patience,
                                                               threshold=args.early_stopping_threshold)
    return model, optimizer, scheduler


def determine_largest_batch_size(args, data, device, angle_means, fraction_to_keep=0.8, headroom=0.9, max_probe_s=300.0):
    """`-adbs / --automatically_determine_batch_size` (train.py:532-551 + scripts/determine_largest_batchsize.py:19-110): the
    largest `-b` whose training batches - drawn from the LARGEST length bin with the residue budget b x max_seq_len, as the
    reference's probe draws them (`use_largest_bin=True`) - train on this GPU, times `fraction_to_keep`.

    The reference doubles b in a fresh subprocess per attempt until CUDA runs out of memory (the subprocess exists because
    a crashed CUDA context keeps its cache).  Here the probe stays in the process and never drives the device out of memory:
    two `train_step`s per candidate on a scratch model, the peak of the caching allocator measured around them, and the
    doubling stops when the NEXT candidate's extrapolated peak (linear in the batch's tokens, fitted to the two most recent
    probes) would exceed `headroom` x the memory that is free (`torch.cuda.mem_get_info`) - or when a candidate does raise
    an out-of-memory error (then the last b that trained stands, as in the reference), when the budget covers the whole
    training set, or after `max_probe_s` seconds.  Under data parallelism every rank probes its own share of the batches and
    the ranks agree on the smallest answer."""
    from math import ceil
    from .dataset import BinnedProteinDataset, SimilarLengthBatchSampler, make_paired_collate_fn
    t_start = time.time()
    world = dp.world_size()
    max_seq_len = getattr(args, "max_seq_len", MAX_SEQ_LEN)
    ds = BinnedProteinDataset(seqs=data['train']['seq'], crds=data['train']['crd'], angs=data['train']['ang'],
                              add_sos_eos=args.add_sos_eos, skip_missing_residues=args.skip_missing_res_train,
                              bins=args.bins, max_seq_len=max_seq_len)
    collate = make_paired_collate_fn(max_seq_len)
    total_tokens = int(sum(min(int(n), max_seq_len) for n in ds.lens))
    b_cap = max(1, ceil(total_tokens / max_seq_len))            # a budget beyond the whole training set changes nothing
    if dp.is_main():
        print("Determining maximum batch size.")

    def probe(b):
        """Two training steps on a largest-bin batch with the budget of `b`; (tokens of this rank's share, allocator peak)."""
        smp = SimilarLengthBatchSampler(ds, b, dynamic_batch=b * max_seq_len, optimize_batch_for_cpus=False,
                                        use_largest_bin=True)
        idx = next(iter(smp))
        idx = list(idx)[dp.rank()::world] if world > 1 else list(idx)
        seq, ang, crd = collate([ds[int(i)] for i in idx]) if idx else (torch.zeros(0, 1, dtype=torch.long),) * 3
        model = optimizer = None
        torch.cuda.synchronize(device)
        torch.cuda.reset_peak_memory_stats(device)
        try:
            model, optimizer, _ = setup_model_optimizer_scheduler(args, device, angle_means)
            seq, ang, crd = seq.to(device), ang.to(device), crd.to(device)
            for _ in range(2):
                if idx:
                    with dp.single_process(model):          # (no collective inside the probe: the ranks may stop at different b)
                        train_step(model, optimizer, args, seq, ang, crd, n_res=int((seq != VOCAB.pad_id).sum()))
            torch.cuda.synchronize(device)
            peak = torch.cuda.max_memory_allocated(device)
        finally:
            del model, optimizer, seq, ang, crd
            from . import _lib
            _lib._workspaces.clear()
            torch.cuda.empty_cache()
        return int(len(idx) * (max(min(int(ds.lens[int(i)]), max_seq_len) for i in idx) if idx else 0)), int(peak)

    np_state, torch_state = np.random.get_state(), torch.get_rng_state()      # the probe must not move the run's RNG streams
    b, best, history, why = 1, 0, [], "budget covers the training set"
    try:
        while True:
            try:
                tokens, peak = probe(b)
            except (torch.cuda.OutOfMemoryError, RuntimeError) as e:
                if "out of memory" not in str(e).lower() and not isinstance(e, torch.cuda.OutOfMemoryError):
                    raise
                torch.cuda.empty_cache()
                why = f"b = {b} ran out of memory"
                break
            best = b
            history.append((b, tokens, peak))
            if dp.is_main():
                print(f"Testing batch size {b: >4}: {tokens} tokens on this GPU, peak {peak / 2 ** 30:.2f} GiB - success.")
            if b >= b_cap:
                break
            if time.time() - t_start > max_probe_s:
                why = f"probe time limit ({max_probe_s:.0f} s)"
                break
            nxt = min(2 * b, b_cap)
            free, _total = torch.cuda.mem_get_info(device)
            if len(history) >= 2 and history[-1][1] > history[-2][1]:
                (b0, t0, p0), (b1, t1, p1) = history[-2], history[-1]
                per_token = max(0.0, (p1 - p0) / (t1 - t0))
                predicted = p1 + per_token * (t1 * nxt / b1 - t1)
                if predicted > headroom * (free + p1):     # (free excludes what the probe itself held and has returned)
                    # the largest b the fit allows, tried once more if it is a real step beyond the last success
                    fit = int(b1 + (headroom * (free + p1) - p1) / max(per_token * t1 / b1, 1.0))
                    if fit >= b1 + max(1, b1 // 10) and fit < nxt:
                        nxt = fit
                    else:
                        why = f"b = {nxt} would need {predicted / 2 ** 30:.1f} GiB of the {free / 2 ** 30:.1f} GiB free"
                        break
            b = nxt
    finally:
        np.random.set_state(np_state)
        torch.set_rng_state(torch_state)
    best = max(1, best)
    if world > 1:
        t = torch.tensor([best], dtype=torch.int64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
        best = int(t.item())
    # train.py:545-548: the reference keeps b itself when the CPU loss workers bound it and 80 % of it otherwise; there are
    # no CPU workers here, so it is always the memory that speaks
    max_batch_size = max(1, ceil(best * fraction_to_keep))
    if dp.is_main():
        print(f"Maximum batch size found to be {best} ({why}). Will proceed with {max_batch_size}. "
              f"{int((time.time() - t_start) // 60)} min elapsed.")
    return max_batch_size


def create_parser():
    """The reference's argument parser (train.py:396-529), flag for flag."""
    def my_bool(s):
        return s != 'False'
    parser = argparse.ArgumentParser()

    required = parser.add_argument_group("Required Args")
    required.add_argument('--data', help="Path to training data.", default="../data/proteinnet/casp12_200123_30.pt")
    required.add_argument("--name", type=str, help="The model name.", default=None)

    training = parser.add_argument_group("Training Args")
    training.add_argument("-lr", "--learning_rate", type=float, default=1 * (10 ** -4))
    training.add_argument('-e', '--epochs', type=int, default=10)
    training.add_argument("-b", '--batch_size', type=int, default=8)
    training.add_argument('-es', '--early_stopping', type=int, default=20)
    training.add_argument('-nws', '--n_warmup_steps', type=int, default=10_000)
    training.add_argument('-cg', '--clip', type=float, default=1)
    training.add_argument('-l', '--loss', choices=["mse", "drmsd", "lndrmsd", "combined"], default="combined")
    training.add_argument('--train_only', action='store_true')
    training.add_argument('--lr_scheduling', type=str, choices=['noam', 'plateau'], default='plateau')
    training.add_argument('--patience', type=int, default=10)
    training.add_argument('--early_stopping_threshold', type=float, default=0.001)
    training.add_argument('-esm', '--early_stopping_metric', type=str, default=None,
                          help="<train|test|valid-NN>-<mse|drmsd|lndrmsd|combined>")
    training.add_argument('--without_angle_means', action='store_true')
    training.add_argument('--eval_train', type=my_bool, default=False)
    training.add_argument('-opt', '--optimizer', type=str, choices=['adam', 'sgd'], default='sgd')
    training.add_argument("-fctf", "--fraction_complete_tf", type=float, default=1)
    training.add_argument("-fsstf", "--fraction_subseq_tf", type=float, default=1)
    training.add_argument("--skip_missing_res_train", type=my_bool, default=False)
    training.add_argument("--repeat_train", type=int, default=1)
    training.add_argument("-s", "--seed", type=int, default=11_731)
    training.add_argument("--combined_drmsd_weight", type=float, default=0.5)
    training.add_argument("--batching_order", type=str, choices=["descending", "ascending", "binned-random"],
                          default="binned-random")
    training.add_argument('--backbone_loss', action='store_true')
    training.add_argument('--sequential_drmsd_loss', action="store_true")
    training.add_argument("--bins", type=int, default=-1)
    training.add_argument("--train_eval_downsample", type=float, default=.10)
    training.add_argument("-adbs", "--automatically_determine_batch_size", type=my_bool, default=False)

    model_args = parser.add_argument_group("Model Args")
    model_args.add_argument('-m', '--model', type=str, default="enc-only")
    model_args.add_argument('-dm', '--d_model', type=int, default=512)
    model_args.add_argument('-dih', '--d_inner_hid', type=int, default=2048)
    model_args.add_argument('-nh', '--n_head', type=int, default=8)
    model_args.add_argument('-nl', '--n_layers', type=int, default=6)
    model_args.add_argument('-do', '--dropout', type=float, default=0.1)
    model_args.add_argument('--postnorm', action='store_true')
    model_args.add_argument("--weight_decay", type=my_bool, default="True")
    model_args.add_argument("--conv1_size", type=int, default=None)
    model_args.add_argument("--conv2_size", type=int, default=None)
    model_args.add_argument("--conv3_size", type=int, default=None)
    model_args.add_argument("--conv1_reduc", type=int, default=None)
    model_args.add_argument("--conv2_reduc", type=int, default=None)
    model_args.add_argument("--conv3_reduc", type=int, default=None)
    model_args.add_argument("--use_embedding", type=my_bool, default="True")
    model_args.add_argument("--conv_out_matches_dm", type=my_bool, default="True")

    saving_args = parser.add_argument_group("Saving Args")
    saving_args.add_argument('--log_structure_step', type=int, default=10)
    saving_args.add_argument('--structure_dir', type=str, default=None,
                             help='write <step>_pred.pdb / true.pdb of the first protein of a batch there every '
                                  '--log_structure_step steps (off when not given: it costs a device sync)')
    saving_args.add_argument('-lvs', '--log_val_struct_step', type=int, default=50)
    saving_args.add_argument('--log_wandb_step', type=int, default=1)
    saving_args.add_argument("-png", '--save_pngs', type=my_bool, default=True)
    saving_args.add_argument('--no_cuda', action='store_true')
    saving_args.add_argument('-c', '--cluster', type=my_bool, default=False)
    saving_args.add_argument('--restart', action='store_true')
    saving_args.add_argument('--restart_opt', action='store_true')
    saving_args.add_argument("--checkpoint_time_interval", type=float, default=0)
    saving_args.add_argument("--load_chkpt", type=str, default=None)

    new = parser.add_argument_group("MI355X path additions (not in the reference)")
    new.add_argument("--max_seq_len", type=int, default=MAX_SEQ_LEN,
                     help="Positional table size / truncation length (the reference hard-wires 500).")
    new.add_argument("--synthetic", type=str, default=None,
                     help="'B,L[,n_batches]': train on generated fixed-length batches instead of --data.")
    new.add_argument("--log_dir", type=str, default="../data/logs")
    new.add_argument("--chkpt_dir", type=str, default="../data/checkpoints")
    new.add_argument("--reference-csv", dest="reference_csv", action="store_true",
                     help="Write the granularity column of the .train log exactly like the reference (the literal 'epoch' "
                          "on per-batch rows too, log.py:130) instead of 'batch' / 'epoch'.")
    return parser


def main():
    """train.py:553-676 without the wandb / PyMOL requirements."""
    global START_TIME
    parser = create_parser()
    args = parser.parse_args()
    if args.no_cuda or not torch.cuda.is_available():
        sys.exit("protein_transformer_amd runs on the MI355X only; use the reference itself for --no_cuda runs.")
    assert args.name is None or "_" not in args.name, "Please do not use underscores in experiment names."   # :577
    args.cuda = True
    args.es_mode, args.es_metric = (args.early_stopping_metric or f"train-{args.loss}").rsplit("-", 1)
    args.add_sos_eos = args.model == "enc-dec"
    args.bins = "auto" if args.bins == -1 else args.bins
    if "conv-enc" in args.model:
        ks, rs = parse_conv_kernel_info_from_model_name(args.model)
        for i, (k, r) in enumerate(zip(ks, rs), 1):
            setattr(args, f"conv{i}_size", k)
            setattr(args, f"conv{i}_reduc", r)
        args.model = args.model.split("|")[0]
    dp.init_from_env()
    device = torch.device("cuda", dp.local_rank())
    torch.cuda.set_device(device)
    drmsd_worker_pool = init_worker_pool(args)
    seed_rngs(args)

    if args.synthetic:
        from .synthetic_data import make_synthetic_dataset
        data = make_synthetic_dataset(args.synthetic, args.seed, device)
    else:
        data = torch.load(args.data, weights_only=False)
    args.max_len = data["settings"]["max_len"]
    angle_means = data["settings"]["angle_means"]
    if args.automatically_determine_batch_size:                   # train.py:586-587
        args.batch_size = determine_largest_batch_size(args, data, device, angle_means)
        seed_rngs(args)                                           # the run starts from the seeds it would have had without the probe

    model, optimizer, scheduler = setup_model_optimizer_scheduler(args, device, angle_means)
    args.name = args.name or time.strftime("run%y%m%d-%H%M%S")
    os.makedirs(args.log_dir, exist_ok=True)
    os.makedirs(args.chkpt_dir, exist_ok=True)
    args.log_file = os.path.join(args.log_dir, args.name + '.train')
    from . import log as _log
    _log.REFERENCE_CSV = bool(args.reference_csv)
    args.chkpt_path = os.path.join(args.chkpt_dir, args.name)
    START_TIME = time.time()                      # load_model moves it back by the checkpoint's elapsed time
    model, optimizer, scheduler, resumed, metrics = load_model(model, optimizer, scheduler, args)
    dp.attach(model)                              # per-layer gradient all-reduce overlapped with backward (no-op for 1 rank)
    log_f = open(args.log_file, 'a' if resumed else 'w', buffering=1) if dp.is_main() else open(os.devnull, "w")
    log_writer = csv.writer(log_f)
    if not resumed:
        log_writer.writerow(prepare_log_header(args).split(","))
    training_data, training_eval_loader, validation_datasets, test_data = prepare_dataloaders(data, args, args.max_seq_len)
    train(model, metrics, training_data, training_eval_loader, validation_datasets, test_data, optimizer, device, args,
          log_writer, scheduler, drmsd_worker_pool)
    log_f.close()
    dp.shutdown()


if __name__ == '__main__':
    main()
