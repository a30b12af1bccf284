"""Datasets, length-binned batching and collation with the reference's API and on-disk format.

Mirrors /root/reference/protein_transformer/dataset.py: `paired_collate_fn` :13-23, `collate_fn` :26-54,
`ProteinDataset` :57-100, `BinnedProteinDataset` :103-158, `SimilarLengthBatchSampler` :161-225,
`prepare_dataloaders` :228-290.  Input: the `torch.save`d dictionary described in SURVEY.md Appendix D
(`data[split]["seq"|"ang"|"crd"]`, `data["settings"]["angle_means"|"max_len"]`).

Differences by design: the truncation length is a parameter (`max_seq_len`, default 500 = the
reference's hard-wired MAX_SEQ_LEN) and batches are pinned so the H2D copy overlaps compute.  The batch
arithmetic (bins, per-bin batch size = residue budget / bin edge, sampling with replacement, rounding
to a multiple of the CPU count for dRMSD losses) is reproduced exactly, including the `"ln-drmsd"`
spelling that never matches `lndrmsd` (dataset.py:250-251).
"""
import numpy as np
import torch
import torch.utils.data

from .protein.Sequence import VOCAB, ProteinVocabulary
from .protein.Structure import NUM_PREDICTED_COORDS

VALID_SPLITS = [10, 20, 30, 40, 50, 70, 90]
MAX_SEQ_LEN = 500


def collate_fn(insts, coords=False, sequences=False, max_seq_len=None, min_len=0):
    """Pad every instance to the longest in the batch (pad id 20 for sequences, zeros otherwise), then
    truncate to max_seq_len residues (x14 atoms for coordinates).  `min_len` (residues): pad at least that far - a
    data-parallel shard is padded to the longest protein of the GLOBAL batch (ShardedBatchSampler)."""
    longest = max(max(len(inst) for inst in insts), min_len * (NUM_PREDICTED_COORDS if coords else 1))
    rows = []
    for inst in insts:
        inst = np.asarray(inst)
        if sequences:
            pad = np.ones((longest - len(inst))) * VOCAB.pad_id
        else:
            pad = np.zeros((longest - len(inst), inst.shape[-1]))
        rows.append(np.concatenate((inst, pad), axis=0))
    batch = np.array(rows)
    batch = batch[:, :max_seq_len * NUM_PREDICTED_COORDS] if coords else batch[:, :max_seq_len]
    return torch.LongTensor(batch) if sequences else torch.FloatTensor(batch)


def pack_batch(tensors, pin=False):
    """The tensors of a batch as views of ONE host buffer (sections aligned to 64 bytes), so that the batch goes to the
    device as one copy instead of one per tensor (DevicePrefetcher; profiles/r05/r05_upload_cost.txt: the step with three
    copies on the side stream costs +0.03 ... 0.05 ms over resident batches, with one copy nothing).  Same values, shapes and
    dtypes; `pin`: allocate the buffer in page-locked memory (main process only)."""
    tensors = [t.contiguous() for t in tensors]
    offs, n = [], 0
    for t in tensors:
        offs.append(n)
        n += (t.numel() * t.element_size() + 63) // 64 * 64
    buf = torch.empty(max(n, 64), dtype=torch.uint8, pin_memory=bool(pin))
    views = []
    for t, o in zip(tensors, offs):
        v = buf[o:o + t.numel() * t.element_size()].view(t.dtype).view(t.shape)
        v.copy_(t)
        views.append(v)
    return tuple(views)


def packed_base(tensors):
    """The single byte buffer behind a batch made by `pack_batch` (None for any other batch)."""
    if not tensors or any(not torch.is_tensor(t) or t.device.type != "cpu" or not t.is_contiguous() for t in tensors):
        return None
    st = tensors[0].untyped_storage()
    if st.nbytes() == 0 or any(t.untyped_storage().data_ptr() != st.data_ptr() for t in tensors[1:]):
        return None
    return torch.empty(0, dtype=torch.uint8).set_(st)


def make_paired_collate_fn(max_seq_len=MAX_SEQ_LEN, packed=True):
    def paired(insts):
        if len(insts) == 0:      # this rank's shard of a batch with fewer proteins than ranks (dp.shard_indices)
            return (torch.zeros(0, 0, dtype=torch.int64), torch.zeros(0, 0, 24), torch.zeros(0, 0, 3))
        fields = list(zip(*insts))
        sequences, angles, coords = fields[:3]
        pad_to = max(fields[3]) if len(fields) > 3 else 0       # the global batch's longest protein (data parallel)
        batch = (collate_fn(sequences, sequences=True, max_seq_len=max_seq_len, min_len=pad_to),
                 collate_fn(angles, max_seq_len=max_seq_len, min_len=pad_to),
                 collate_fn(coords, coords=True, max_seq_len=max_seq_len, min_len=pad_to))
        return pack_batch(batch) if packed else batch
    return paired


paired_collate_fn = make_paired_collate_fn(MAX_SEQ_LEN)


def _load(seqs, angs, crds, add_sos_eos, skip_missing_residues):
    assert seqs is not None
    assert (angs is None) or (len(seqs) == len(angs) and len(angs) == len(crds))
    s, a, c = [], [], []
    for i in range(len(seqs)):
        if skip_missing_residues and np.isnan(angs[i]).all(axis=-1).any():
            continue
        s.append(VOCAB.str2ints(seqs[i], add_sos_eos))
        a.append(angs[i])
        c.append(crds[i])
    return s, a, c


class ProteinDataset(torch.utils.data.Dataset):
    """Sequences, angles and coordinates of a split, optionally sorted by length (longest first)."""

    def __init__(self, seqs=None, angs=None, crds=None, add_sos_eos=True, sort_by_length=True, reverse_sort=True,
                 skip_missing_residues=True):
        self._seqs, self._angs, self._crds = _load(seqs, angs, crds, add_sos_eos, skip_missing_residues)
        if sort_by_length:
            order = [i for i, _ in sorted(enumerate(self._angs), key=lambda x: x[1].shape[0], reverse=reverse_sort)]
            self._seqs = [self._seqs[i] for i in order]
            self._angs = [self._angs[i] for i in order]
            self._crds = [self._crds[i] for i in order]

    @property
    def n_insts(self):
        return len(self._seqs)

    def __len__(self):
        return self.n_insts

    def __getitem__(self, idx):
        if isinstance(idx, tuple):               # (index, pad_to) from ShardedBatchSampler: the length travels with the item
            return (*self[idx[0]], idx[1])
        if self._angs is not None:
            return self._seqs[idx], self._angs[idx], self._crds[idx]
        return self._seqs[idx]


class BinnedProteinDataset(torch.utils.data.Dataset):
    """Like ProteinDataset, plus a length histogram; assumes the data is sorted shortest to longest."""

    def __init__(self, seqs=None, angs=None, crds=None, add_sos_eos=True, skip_missing_residues=True, bins="auto",
                 max_seq_len=MAX_SEQ_LEN):
        self.vocab = ProteinVocabulary()
        self._seqs, self._angs, self._crds = _load(seqs, angs, crds, add_sos_eos, skip_missing_residues)
        self.lens = [min(len(x), max_seq_len) for x in self._seqs]
        self.hist_counts, edges = np.histogram(self.lens, bins=bins)
        self.hist_bins = edges[1:]                       # right edge of every bin: '( , ]'
        self.bin_probs = self.hist_counts / self.hist_counts.sum()
        self.bin_map = {}
        seq_i = bin_j = 0
        while seq_i < len(self._seqs):
            if self.lens[seq_i] <= self.hist_bins[bin_j]:
                self.bin_map.setdefault(bin_j, []).append(seq_i)
                seq_i += 1
            else:
                bin_j += 1

    @property
    def n_insts(self):
        return len(self._seqs)

    def __len__(self):
        return self.n_insts

    def __getitem__(self, idx):
        if isinstance(idx, tuple):               # (index, pad_to) from ShardedBatchSampler
            return (*self[idx[0]], idx[1])
        if self._angs is not None:
            return self._seqs[idx], self._angs[idx], self._crds[idx]
        return self._seqs[idx]


class SimilarLengthBatchSampler(torch.utils.data.Sampler):
    """Yields index batches drawn (with replacement) from one random length bin at a time; with
    `dynamic_batch` the batch holds about that many residues."""

    def __init__(self, data_source, batch_size, dynamic_batch, optimize_batch_for_cpus, downsample=None,
                 use_largest_bin=False, repeat_train=None):
        self.data_source = data_source
        self.batch_size = batch_size
        self.dynamic_batch = dynamic_batch
        self.optimize_batch_for_cpus = optimize_batch_for_cpus
        # the reference rounds dRMSD batches down to a multiple of its CPU loss workers (dataset.py:186,218-220); the
        # loss workers of this path are the GPUs of the job, which `prepare_dataloaders` writes here (1 GPU: no rounding)
        self.cpu_count = torch.multiprocessing.cpu_count()
        self.min_batch = 1
        self.downsample = downsample
        self.use_largest_bin = use_largest_bin
        self.repeat_train = repeat_train if repeat_train else 1

    def __len__(self):
        if self.dynamic_batch:
            numerator, divisor = sum(self.data_source.lens) * self.repeat_train, self.dynamic_batch
        else:
            numerator, divisor = len(self.data_source) * self.repeat_train, self.batch_size
        if self.downsample:
            numerator *= self.downsample
        return int(np.ceil(numerator / divisor))

    def __iter__(self):
        ds = self.data_source
        for _ in range(len(self)):
            if self.use_largest_bin:
                b = len(ds.hist_bins) - 1
            else:
                b = np.random.choice(range(len(ds.hist_bins)), p=ds.bin_probs)
            if self.dynamic_batch:
                size = int(self.dynamic_batch / ds.hist_bins[b])
                if self.optimize_batch_for_cpus:
                    size -= size % self.cpu_count
                size = max(self.min_batch, size)
            else:
                size = self.batch_size
            yield np.random.choice(ds.bin_map[b], size=size)


_SIDE_STREAMS = {}


class DevicePrefetcher:
    """Host -> device hand-over of the batches of a loader, one batch ahead: while the step of batch i runs on the
    compute stream, batch i + 1 is collated by the loader's worker, counted (non-pad residues, on the host) and copied
    from pinned memory on a side stream; the compute stream only waits for the copy's event.  Yields
    (seq, ang, crd, n_residues) with the tensors on `device`.  Works for any iterable of (seq, ang, crd) CPU tensors; a batch
    whose tensors are views of one buffer (`pack_batch`: what the collate function of this module returns) travels as ONE
    copy - pinned here if the loader did not - and arrives as views of one device buffer."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        dev = self.device
        side = _SIDE_STREAMS.get(dev)        # one upload stream per device for the life of the process (not one per epoch)
        if side is None:
            side = _SIDE_STREAMS[dev] = torch.cuda.Stream(dev)
        it = iter(self.loader)

        def fetch():
            try:
                batch = next(it)
            except StopIteration:
                return None
            n_res = int((batch[0] != VOCAB.pad_id).sum())
            base = packed_base(batch)
            if base is not None and not base.is_pinned():
                base = base.pin_memory()                     # (the caching host allocator: no hipHostMalloc after the first batches)
            with torch.cuda.stream(side):
                if base is not None:
                    d = base.to(dev, non_blocking=True)
                    on_dev = tuple(d[o:o + t.numel() * t.element_size()].view(t.dtype).view(t.shape)
                                   for t, o in ((t, t.storage_offset() * t.element_size()) for t in batch))
                    owners = (d,)
                else:
                    on_dev = owners = tuple(t.to(dev, non_blocking=True) for t in batch)
                done = torch.cuda.Event()
                done.record(side)
            return on_dev, n_res, done, owners

        nxt = fetch()
        while nxt is not None:
            (seq, ang, crd), n_res, done, owners = nxt
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(done)
            for t in owners:
                t.record_stream(cur)                 # allocated on the side stream, consumed on the compute stream
            nxt = fetch()                            # the next copy is in flight while the caller works on this batch
            yield seq, ang, crd, n_res


class ShardedBatchSampler(torch.utils.data.Sampler):
    """Data-parallel view of a batch sampler: yields, for every batch of the wrapped sampler, the indices that THIS rank
    takes (serpentine deal by length, dp.shard_indices) - so each rank collates, pads and uploads only its shard.  Every
    rank draws the same global batches (same seed, same numpy stream: train.seed_rngs).  With `pad_to_global` (default)
    the indices travel as (index, longest protein of the GLOBAL batch) and the collate function pads that far: a
    conv-enc model then sees the same columns behind a protein's end (pad-token embeddings, not the Conv1d zero padding)
    as in the single-process batch, so the sum of the ranks' gradients is the single-process gradient for every model."""

    def __init__(self, batch_sampler, lengths, world=None, rank=None, pad_to_global=True):
        self.batch_sampler, self.lengths = batch_sampler, lengths
        self.world, self.rank, self.pad_to_global = world, rank, pad_to_global

    def __len__(self):
        return len(self.batch_sampler)

    def __iter__(self):
        from . import dp
        for batch in self.batch_sampler:
            batch = [int(i) for i in batch]
            lens = [int(self.lengths[i]) for i in batch]
            keep = dp.shard_indices(lens, self.world, self.rank)
            yield [(batch[k], max(lens)) for k in keep] if self.pad_to_global else [batch[k] for k in keep]


def prepare_dataloaders(data, args, max_seq_len, num_workers=1):
    """train (binned, dynamic batches), train-eval, 7 validation splits and test loaders (dataset.py:228-290).  With more
    than one rank every loader is sharded (ShardedBatchSampler) and dRMSD batch sizes are multiples of the rank count."""
    from . import dp
    if args.batching_order in ["descending", "ascending"]:
        raise NotImplementedError("Descending and ascending order have not been reimplemented.")
    world = dp.world_size()
    collate = make_paired_collate_fn(max_seq_len)
    cpu_opt = args.loss in ["combined", "drmsd", "ln-drmsd"]
    # (no pin thread: it would pin the three views of a packed batch apart - DevicePrefetcher pins the one buffer)
    common = dict(num_workers=num_workers, collate_fn=collate, pin_memory=False)
    train_dataset = BinnedProteinDataset(seqs=data['train']['seq'], crds=data['train']['crd'], angs=data['train']['ang'],
                                         add_sos_eos=args.add_sos_eos, skip_missing_residues=args.skip_missing_res_train,
                                         bins=args.bins, max_seq_len=max_seq_len)

    def sharded(sampler, lengths):
        return ShardedBatchSampler(sampler, lengths) if world > 1 else sampler

    def binned_sampler(**kw):
        smp = SimilarLengthBatchSampler(train_dataset, args.batch_size, optimize_batch_for_cpus=cpu_opt, **kw)
        smp.cpu_count = world            # the "loss workers" of this path are the GPUs; every rank gets a protein
        smp.min_batch = world if cpu_opt else 1
        return sharded(smp, train_dataset.lens)

    train_loader = torch.utils.data.DataLoader(
        train_dataset, batch_sampler=binned_sampler(dynamic_batch=args.batch_size * max_seq_len,
                                                    repeat_train=args.repeat_train), **common)
    train_eval_loader = torch.utils.data.DataLoader(
        train_dataset, batch_sampler=binned_sampler(dynamic_batch=None, downsample=args.train_eval_downsample), **common)

    def plain(split):
        ds = ProteinDataset(seqs=data[split]['seq'], crds=data[split]['crd'], angs=data[split]['ang'],
                            add_sos_eos=args.add_sos_eos, skip_missing_residues=args.skip_missing_res_train)
        if world == 1:
            return torch.utils.data.DataLoader(ds, batch_size=args.batch_size, **common)
        batches = torch.utils.data.BatchSampler(torch.utils.data.SequentialSampler(ds), args.batch_size, drop_last=False)
        lengths = [min(len(q), max_seq_len) for q in ds._seqs]      # from the raw sequences, no __getitem__ per item
        return torch.utils.data.DataLoader(ds, batch_sampler=sharded(batches, lengths), **common)

    valid_loaders = {split: plain(f'valid-{split}') for split in VALID_SPLITS if f'valid-{split}' in data}
    test_loader = plain('test') if 'test' in data else None
    return train_loader, train_eval_loader, valid_loaders, test_loader
