"""Metrics bookkeeping, residues/s speed meter and the CSV `.train` log of the training driver.

Keeps the metric-dictionary keys, the CSV columns and the speed definition of
/root/reference/protein_transformer/log.py (`update_loss_trackers` :92-112, `log_batch` :115-130,
`init_metrics` :359-386, `update_metrics` :389-434 with speed = non-pad residues / time since the
previous batch :422-430, `reset_metrics_for_epoch` :437-457, `update_metrics_end_of_epoch` :460-485,
`prepare_log_header` :488-495, `EarlyStoppingCondition` :498-503) and the PDB part of its structure dumps
(`log_structure_and_angs` :310-337, without wandb / PyMOL).  wandb and tqdm status bars are left out.
"""
import sys
import time

import numpy as np

from . import dp
from .dataset import VALID_SPLITS
from .protein.Sequence import VOCAB

_TRACKED = ("drmsd-full", "lndrmsd-full", "mse-full", "combined-full", "rmsd-full", "drmsd-bb", "lndrmsd-bb",
            "mse-bb", "mse-sc")
_HISTORY = ("drmsd", "combined", "lndrmsd", "mse")


class EarlyStoppingCondition(Exception):
    """Raised when the early-stopping patience is exhausted."""


def _num(x):
    if x is None:
        return 0.0
    return float(x.item()) if hasattr(x, "item") else float(x)


def _split_metrics():
    return {f"epoch-history-{h}": [] for h in _HISTORY}


def init_metrics(args):
    metrics = {"train": _split_metrics(), "test": _split_metrics(), "history-lr": [], "epoch_last_improved": -1,
               "best_valid_loss_so_far": np.inf, "last_chkpt_time": time.time(), "n_batches": 0}
    for split in VALID_SPLITS:
        metrics[f"valid-{split}"] = _split_metrics()
    if args.lr_scheduling != "noam":
        metrics["history-lr"] = [0]
    return metrics


def reset_metrics_for_epoch(metrics, mode):
    m = metrics.setdefault(mode, _split_metrics())
    for k in _TRACKED:
        m[f"epoch-{k}"] = m[f"batch-{k}"] = 0
    m["batch-history"], m["speed-history"] = [], []
    m["batch-time"] = time.time()
    metrics["n_batches"] = 0
    return metrics


def update_metrics(metrics, losses, mode, src_seq, tracking_loss=None, batch_level=True):
    m = metrics[mode]
    if batch_level:
        metrics["n_batches"] += 1
    for k in _TRACKED:
        v = _num(losses[k])
        if batch_level:
            m[f"batch-{k}"] = v
        if k.endswith("-full"):
            m[f"epoch-{k}"] += v
        else:
            m[f"epoch-{k}"] = v                       # the reference overwrites the bb/sc entries (log.py:413-416)
    # residues of the GLOBAL batch: counted on the host before the upload and reduced with the loss statistics
    # (losses.LossReport); the device count is only the fall-back for callers that did not pass one
    num_res = losses.get("n-residues")
    if num_res is None:
        num_res = int((src_seq != VOCAB.pad_id).sum().item()) * dp.world_size()
    now = time.time()
    m["speed"] = num_res / max(now - m["batch-time"], 1e-9)        # log.py:422-424
    m.setdefault("speeds", []).append(m["speed"])
    m["batch-time"] = now
    m["speed-history"].append(m["speed"])
    if tracking_loss is not None:
        m["batch-history"].append(float(tracking_loss))
    return metrics


def update_metrics_end_of_epoch(metrics, mode):
    n = max(metrics["n_batches"], 1)
    m = metrics[mode]
    for k in ("drmsd-full", "lndrmsd-full", "mse-full", "drmsd-bb", "lndrmsd-bb", "mse-bb", "mse-sc", "rmsd-full"):
        m[f"epoch-{k}"] /= n
    m["epoch-combined-full"] = 0 if m["epoch-drmsd-full"] == 0 else m["epoch-combined-full"] / n
    for h in _HISTORY:
        m[f"epoch-history-{h}"].append(m[f"epoch-{h}-full"])
    return metrics


def update_loss_trackers(args, epoch_i, metrics):
    loss_to_compare = metrics[args.es_mode][f"epoch-{args.es_metric}-full"]
    losses_to_compare = metrics[args.es_mode][f"epoch-history-{args.es_metric}"]
    if metrics["best_valid_loss_so_far"] - loss_to_compare > args.early_stopping_threshold:
        metrics["best_valid_loss_so_far"] = loss_to_compare
        metrics["epoch_last_improved"] = epoch_i
    elif args.early_stopping and epoch_i - metrics["epoch_last_improved"] > args.early_stopping:
        print("No improvement for {} epochs. Stopping model training early.".format(args.early_stopping))
        raise EarlyStoppingCondition
    metrics["loss_to_compare"] = loss_to_compare
    metrics["losses_to_compare"] = losses_to_compare
    return metrics


REFERENCE_CSV = False     # True: the granularity column carries upstream's literal "epoch" on every row (log.py:130)


def prepare_log_header(args):
    if args.loss == "combined":
        return 'drmsd,ln_drmsd,rmse,rmsd,combined,lr,mode,granularity,time,speed'
    return 'drmsd,ln_drmsd,rmse,rmsd,lr,mode,granularity,time,speed'


def log_batch(log_writer, metrics, start_time, mode="valid", end_of_epoch=False, t=None):
    """One CSV row (log.py:115-130): ten values - drmsd, ln_drmsd, rmse, rmsd, combined, lr, mode, granularity, time,
    speed; like upstream the `combined` value is written whatever the header of `prepare_log_header` lists.  One
    deliberate difference: the granularity column says "batch" for per-batch rows (upstream writes the literal "epoch"
    in both cases, log.py:130) - unless `REFERENCE_CSV` is set (`train.py --reference-csv`), which reproduces the
    upstream column byte for byte for consumers of the reference's `.train` files."""
    t = t or time.time()
    m = metrics[mode]
    be = "epoch" if end_of_epoch else "batch"
    label = "epoch" if REFERENCE_CSV else be
    lr = metrics["history-lr"][-1] if metrics["history-lr"] else 0
    log_writer.writerow([m[f"{be}-drmsd-full"], m[f"{be}-lndrmsd-full"], np.sqrt(m[f"{be}-mse-full"]),
                         m[f"{be}-rmsd-full"], m[f"{be}-combined-full"], lr, mode, label, round(t - start_time, 4),
                         m.get("speed", 0)])


def do_train_batch_logging(metrics, losses, src_seq, optimizer, args, log_writer, start_time, step):
    metrics = update_metrics(metrics, losses, "train", src_seq, tracking_loss=_num(losses["loss"]))
    if not np.isfinite(_num(losses["loss"])):
        print("A nan loss has occurred. Exiting training.")          # log.py:182-185
        sys.exit(1)
    lr = optimizer.param_groups[0]["lr"]
    metrics["history-lr"].append(lr)
    if dp.is_main():
        log_batch(log_writer, metrics, start_time, mode="train", end_of_epoch=False)
        if step % max(1, getattr(args, "log_wandb_step", 1) * 10) == 0:
            m = metrics["train"]
            print(f"  step {step:5d}  drmsd {m['batch-drmsd-full']:.4f}  ln {m['batch-lndrmsd-full']:.6f}  "
                  f"rmse {np.sqrt(m['batch-mse-full']):.4f}  comb {m['batch-combined-full']:.4f}  "
                  f"lr {lr:.2e}  {m['speed']:.0f} res/s", flush=True)
    return metrics


def log_structure(args, pred_coords, true_coords, src_seq, step, struct_name="train"):
    """PDB dump of one predicted structure next to its target (log.py:310-337 of the reference without wandb and
    PyMOL): `<structure_dir>/<struct_name>/<step:05>_pred.pdb` and `true.pdb`.

    pred_coords, true_coords: [L*14, 3] of ONE protein (true may carry NaN / zero rows for missing atoms and batch
    padding rows, which are dropped like in the reference), src_seq: [L] residue ids without padding.
    """
    import os

    import torch

    from .protein.PDB_Creator import PDB_Creator
    seq = VOCAB.ints2str([int(i) for i in torch.as_tensor(src_seq).cpu().tolist()])
    path = os.path.join(args.structure_dir, struct_name)
    os.makedirs(path, exist_ok=True)
    pred = torch.as_tensor(pred_coords).detach().cpu().float().numpy()[:len(seq) * 14]
    true = torch.as_tensor(true_coords).detach().cpu().float().clone()[:len(seq) * 14]
    true[torch.isnan(true)] = 0
    pred_path = os.path.join(path, f"{step:05}_pred.pdb")
    PDB_Creator(pred, seq=seq).save_pdb(pred_path, title="pred")
    true_path = os.path.join(path, "true.pdb")
    if not os.path.isfile(true_path) or struct_name == "train":
        PDB_Creator(true.numpy(), seq=seq).save_pdb(true_path, title="true")
    return pred_path, true_path


def do_eval_batch_logging(metrics, losses, src_seq, args, mode):
    return update_metrics(metrics, losses, mode, src_seq, batch_level=False)


def do_eval_epoch_logging(metrics, mode):
    metrics["n_batches"] = max(1, len(metrics[mode].get("speed-history", [])))
    update_metrics_end_of_epoch(metrics, mode)
    if dp.is_main():
        m = metrics[mode]
        print(f"  [{mode}] drmsd {m['epoch-drmsd-full']:.4f}  ln {m['epoch-lndrmsd-full']:.6f}  "
              f"rmse {np.sqrt(m['epoch-mse-full']):.4f}  rmsd {m['epoch-rmsd-full']:.4f}", flush=True)
