"""Loaders -- host-side mirror of the ``TripletDataset`` / ``SeqDataset`` contract.

The batches these loaders hand to ``BaseRetriever`` are the interface to keep
(recstudio/data/dataset.py): training batches are dicts of 1-D tensors
``{user_id, item_id, rating}`` gathered by an index batch (:893-913, :835-868); evaluation
batches are per-user rows ``[user, start, end]`` materialised as padded ``item_id [B,T]``,
``rating [B,T]`` plus ``user_hist [B,Lh]`` (:864-866, :905-906); ``SeqDataset`` batches carry
``in_item_id [B,L]``, ``seqlen [B]`` and the target ``item_id [B]`` (:1418-1439).  The
preprocessing pipeline reproduces the reference's semantics for the interaction file
(rating threshold :179-191, first-occurrence de-duplication :521-526, k-core filter :528-573,
token -> index by first appearance with ``[PAD]`` = 0 :417-474, sort by (user, time) :991-996,
per-user ``np.random.permutation`` shuffle :1001-1005, ratio / leave-one-out split :728-803,
history tables :1165-1189, item frequency :1216-1230) with plain numpy/torch instead of pandas
frames.  Everything here is host code; the device path starts at the batch dict.
"""
import copy
import os
from typing import Dict, Optional

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence

__all__ = ['TripletDataset', 'SeqDataset', 'DataSampler', 'SortedDataSampler', 'RankSlice', 'rank_part',
           'synthetic_interactions']

DEFAULT_CONFIG = {
    'user_id_field': 'user_id:token', 'item_id_field': 'item_id:token', 'rating_field': 'rating:float',
    'time_field': 'timestamp:float', 'inter_feat_name': None, 'inter_feat_header': 0, 'user_feat_name': None,
    'user_feat_header': 0, 'field_separator': '\t', 'min_user_inter': 0, 'min_item_inter': 0,
    'low_rating_thres': None, 'drop_dup': True, 'max_seq_len': 20, 'data_dir': None,
}


def _factorize(values):
    """Index by first appearance (== pandas.factorize): returns (codes, uniques)."""
    uniq, first, inv = np.unique(values, return_index=True, return_inverse=True)
    order = np.argsort(first, kind='stable')
    rank = np.empty_like(order)
    rank[order] = np.arange(len(order))
    return rank[inv], uniq[order]


class DataSampler:
    """recstudio/data/dataset.py:1687-1734: yields index batches (shuffled with a fresh generator
    seeded from torch's global CPU stream, exactly one ``random_()`` draw per epoch)."""

    def __init__(self, data_source, batch_size, shuffle=True, drop_last=False, generator=None):
        self.data_source, self.batch_size = data_source, batch_size
        self.shuffle, self.drop_last, self.generator = shuffle, drop_last, generator

    def __iter__(self):
        n = len(self.data_source)
        if self.generator is None:
            generator = torch.Generator()
            generator.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
        else:
            generator = self.generator
        out = torch.randperm(n, generator=generator) if self.shuffle else torch.arange(n)
        out = out.split(self.batch_size)
        if self.drop_last and len(out[-1]) < self.batch_size:
            out = out[:-1]
        yield from out

    def __len__(self):
        n = len(self.data_source)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size


class SortedDataSampler(DataSampler):
    """recstudio/data/dataset.py:1737-1786: batches of similar sample length (less padding)."""

    def __init__(self, data_source, batch_size, shuffle=False, drop_last=False, generator=None):
        super().__init__(data_source, batch_size, shuffle, drop_last, generator)

    def __iter__(self):
        n = len(self.data_source)
        length = self.data_source.sample_length
        if self.shuffle:
            key = torch.div(torch.randperm(n), self.batch_size * 10, rounding_mode='floor')
            key = length + key * (length.max() + 1)
        else:
            key = length
        out = torch.sort(key).indices.split(self.batch_size)
        if self.drop_last and len(out[-1]) < self.batch_size:
            out = out[:-1]
        yield from out


def _rank_world(rank, world):
    """(rank, world) of this process: the arguments, else torch.distributed, else the launcher's environment."""
    if rank is not None and world is not None:
        return int(rank), int(world)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))


def rank_part(index, rank, world):
    """Rank ``rank``'s CONTIGUOUS 1/world of a global index batch -> (part, n_valid).  A batch the world size does not
    divide is padded by wrapping around to its own first entries (torch's DistributedSampler rule), so that every rank
    gets the same number of rows -- the sharded step's collectives need equal parts; ``n_valid`` says how many rows of
    the part are real (the padding repeats samples: training sees them twice in that one batch, evaluation weighs them
    out)."""
    n = index.shape[0]
    per = (n + world - 1) // world
    pad = per * world - n
    if pad:
        # (a batch shorter than the padding -- n = 3 on 8 ranks needs 5 more rows -- wraps around more than once)
        index = torch.cat([index, index.repeat((pad + n - 1) // n)[:pad]])
    lo = rank * per
    return index[lo:lo + per], max(0, min(per, n - lo))


class RankSlice:
    """recstudio/data/dataset.py:1113-1114, :1141-1142 (``DistributedSamplerWrapper`` around the index-batch sampler) for
    the row-sharded trainer: the wrapped sampler draws GLOBAL index batches -- the same on every rank: it shuffles from
    torch's global CPU stream, which ``seed_everything`` put into the same state everywhere (recommender.py:34-35) -- and
    rank r keeps rows [r * B, (r + 1) * B) of each, so that the ranks' parts concatenated are exactly what ONE process
    running the global batch sees, in that order (together with the job-wide negative stream this makes a G-rank run
    reproduce the single-process run).  The reference deals out whole batches (rank r gets batches r, r + G, ...); a
    contiguous slice of every batch keeps the step's global batch well defined instead.  Yields (index part, n_valid)."""

    def __init__(self, sampler, rank=None, world=None):
        self.sampler = sampler
        self.rank, self.world = _rank_world(rank, world)

    def __iter__(self):
        for index in self.sampler:
            yield rank_part(index, self.rank, self.world)

    def __len__(self):
        return len(self.sampler)


class _Loader:
    """DataLoader(dataset, sampler=index-batch sampler, batch_size=None) equivalent.  With a ``RankSlice`` sampler the
    batch additionally carries ``'_n_valid'`` (a Python int: rows of this rank's part that are not padding)."""

    def __init__(self, dataset, sampler, device=None):
        self.dataset, self.sampler, self.device = dataset, sampler, device

    def __iter__(self):
        for index in self.sampler:
            n_valid = None
            if isinstance(index, tuple):
                index, n_valid = index
            batch = self.dataset[index]
            if self.device is not None:
                batch = {k: v.to(self.device, non_blocking=True) for k, v in batch.items()}
            if n_valid is not None:
                batch['_n_valid'] = n_valid
            yield batch

    def __len__(self):
        return len(self.sampler)


class _DeviceLoader:
    """Loader fast path: the interaction columns and the sample index live on the GPU, an
    epoch's permutation is drawn there, and a batch dict is assembled without touching the host -- for
    SeqDataset the ragged ``[start, end)`` history slices go through ``rsa_seg_gather`` (ids only) instead of
    the per-batch Python ``torch.cat([arange ...])`` + ``pad_sequence`` (dataset.py:1428-1434).  Batches have
    the same keys, dtypes and values as the host loader's."""

    EPOCH_GATHER_BYTES = 8 << 30      # epochs whose permuted columns fit this are gathered once (see __iter__)

    def __init__(self, dataset, batch_size, shuffle, drop_last, device, rank=0, world=1):
        """``world`` > 1: ``batch_size`` is per rank; the epoch's order is drawn for the GLOBAL batches of
        ``batch_size * world`` samples from the device generator (the same state on every rank after ``seed_everything``)
        and this rank keeps its contiguous part of each (see ``RankSlice``); batches carry ``'_n_valid'``."""
        from . import ops                                   # device work only; imported lazily (host-only users)
        self.ops = ops
        self.ds, self.shuffle, self.drop_last = dataset, shuffle, drop_last
        self.rank, self.world, self.per_rank = int(rank), int(world), int(batch_size)
        self.batch_size = int(batch_size) * self.world      # the global batch
        self.device = torch.device(device)
        self.cols = {k: v.to(self.device) for k, v in dataset.inter_feat.items()
                     if k in dataset.use_field or k == dataset.frating}
        self.index = dataset.data_index.to(self.device)
        self.seq = isinstance(dataset, SeqDataset)
        if self.seq:
            self.max_len = int(dataset.sample_length.max())

    def __len__(self):
        n = self.index.shape[0]
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self.index.shape[0]
        if self.seq:      # SortedDataSampler (dataset.py:1767-1780): batches of similar history length
            length = self.index[:, 2] - self.index[:, 1]
            key = length
            if self.shuffle:
                bucket = torch.div(torch.randperm(n, device=self.device), self.batch_size * 10, rounding_mode='floor')
                key = length + bucket * (length.max() + 1)
            order = torch.sort(key, stable=True).indices
            # padded width of every batch of the epoch (the longest history in it) in ONE read-back, instead of an
            # int(lens.max()) per batch -- that sync stalled the stream once per step
            pad = (-n) % self.batch_size
            lens_epoch = torch.cat([length[order], length.new_zeros(pad)]).view(-1, self.batch_size)
            widths = lens_epoch.max(1).values.tolist()
        else:
            order = torch.randperm(n, device=self.device) if self.shuffle else torch.arange(n, device=self.device)
        ds = self.ds
        if not self.seq and self.world == 1 and n * 8 * (len(self.cols) + 1) <= self.EPOCH_GATHER_BYTES:
            # the whole epoch's columns in its order with ONE gather per column; a batch is then a dict of views -- no
            # kernel and no allocation per step (per-batch gathers: 1 + len(cols) launches, a B = 4096 step's worth of time)
            rows = self.index[order]
            epoch = {k: v[rows] for k, v in self.cols.items()}
            del rows
            for lo in range(0, n, self.batch_size):
                if self.drop_last and lo + self.batch_size > n:
                    break
                yield {k: v[lo:lo + self.batch_size] for k, v in epoch.items()}
            return
        for lo in range(0, n, self.batch_size):
            sel = order[lo:lo + self.batch_size]
            if self.drop_last and sel.numel() < self.batch_size:
                break
            n_valid = None
            if self.world > 1:
                sel, n_valid = rank_part(sel, self.rank, self.world)
            rows = self.index[sel]
            if not self.seq:
                batch = {k: v[rows] for k, v in self.cols.items()}
                if n_valid is not None:
                    batch['_n_valid'] = n_valid
                yield batch
                continue
            start, end = rows[:, 1].contiguous(), rows[:, 2].contiguous()
            lens = end - start
            L = int(widths[lo // self.batch_size])
            batch = {ds.fuid: rows[:, 0], 'seqlen': lens}
            ids, _, _ = self.ops.seg_gather(None, self.cols[ds.fiid], start, end, L, want_rows=False)
            batch['in_' + ds.fiid] = ids
            # the CSR view itself, for towers that gather rows straight from it (SASRecQueryEncoder: rsa_seg_gather, rows form)
            batch['_seg'] = (self.cols[ds.fiid], start, end)
            batch[ds.fiid] = self.cols[ds.fiid][end]
            if ds.frating in self.cols:
                # ratings of the history positions: same [start, end) windows, right-padded with 0
                r = self.cols[ds.frating]
                pos = torch.arange(L, device=self.device).view(1, -1) + start.view(-1, 1)
                valid = pos < end.view(-1, 1)
                batch['in_' + ds.frating] = torch.where(valid, r[pos.clamp(max=r.numel() - 1)], torch.zeros((), device=self.device))
                batch[ds.frating] = r[end]
            if n_valid is not None:
                batch['_n_valid'] = n_valid
            yield batch


class TripletDataset:
    def __init__(self, name: str = 'ml-100k', config: Optional[Dict] = None, _interactions=None):
        self.name = name
        self.config = dict(DEFAULT_CONFIG)
        if config:
            self.config.update(config)
        self.fuid = self.config['user_id_field'].split(':')[0]
        self.fiid = self.config['item_id_field'].split(':')[0]
        self.frating = self.config['rating_field'].split(':')[0]
        self.ftime = self.config['time_field'].split(':')[0] if self.config['time_field'] else None
        extra_users = None
        if _interactions is None:
            users, items, ratings, times, extra_users = self._read_atomic_files()
        else:
            users, items, ratings, times = _interactions
        self._preprocess(np.asarray(users), np.asarray(items), ratings, times, extra_users)
        self._use_field = {self.fuid, self.fiid, self.frating}
        self.eval_mode = False

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_interactions(cls, users, items, ratings=None, timestamps=None, config=None, name='synthetic'):
        """Build from raw token arrays (any dtype) instead of atomic files."""
        n = len(users)
        ratings = np.ones(n, dtype=np.float32) if ratings is None else np.asarray(ratings, dtype=np.float64)
        return cls(name, config, _interactions=(users, items, ratings, timestamps))

    @classmethod
    def from_mapped_ids(cls, users, items, ratings=None, n_users=None, n_items=None, config=None, name='synthetic'):
        """Build from id columns that are ALREADY mapped (int64, 1-based, 0 = padding) -- the state ``load_cache`` restores --
        without the token factorisation of ``from_interactions``: synthetic streams and externally preprocessed data at
        sizes where a host-side ``np.unique`` over the stream is the slow part.  ``n_users`` / ``n_items`` are the table sizes
        (ids < n; default max id + 1).  Every interaction is a training sample (``data_index`` = all rows) until ``build``
        splits it."""
        self = cls.__new__(cls)
        self.name = name
        self.config = dict(DEFAULT_CONFIG)
        if config:
            self.config.update(config)
        self.fuid = self.config['user_id_field'].split(':')[0]
        self.fiid = self.config['item_id_field'].split(':')[0]
        self.frating = self.config['rating_field'].split(':')[0]
        self.ftime = self.config['time_field'].split(':')[0] if self.config['time_field'] else None
        users, items = torch.as_tensor(users, dtype=torch.int64), torch.as_tensor(items, dtype=torch.int64)
        if users.shape != items.shape or users.dim() != 1:
            raise ValueError('users and items must be 1-D id columns of the same length')
        self._n_users = int(n_users) if n_users is not None else int(users.max()) + 1
        self._n_items = int(n_items) if n_items is not None else int(items.max()) + 1
        if len(users) and (int(users.max()) >= self._n_users or int(items.max()) >= self._n_items or int(users.min()) < 0
                           or int(items.min()) < 0):
            raise ValueError('mapped ids must lie in [0, n_users) / [0, n_items)')
        ratings = torch.ones(len(users), dtype=torch.float32) if ratings is None else torch.as_tensor(ratings, dtype=torch.float32)
        self.field2tokens = {}
        self.inter_feat = {self.fuid: users, self.fiid: items, self.frating: ratings}
        self.data_index = torch.arange(len(users))
        self._use_field = {self.fuid, self.fiid, self.frating}
        self.eval_mode = False
        return self

    # ------------------------------------------------------------------ flat binary cache
    def save_cache(self, path):
        """The preprocessed dataset as ONE flat binary file (numpy .npz: mapped id columns, ratings, timestamps,
        token tables, config) -- the replacement of the reference's pickled-object cache
        (recstudio/data/dataset.py:95-100, `_save_cache`): loading it skips text parsing, filtering and id
        mapping, and the id columns can be handed to the device loaders as they are."""
        import json
        arrays = {'user': self.inter_feat[self.fuid].numpy(), 'item': self.inter_feat[self.fiid].numpy(),
                  'rating': self.inter_feat[self.frating].numpy(),
                  'user_tokens': self.field2tokens[self.fuid].astype('U'),
                  'item_tokens': self.field2tokens[self.fiid].astype('U'),
                  'meta': np.frombuffer(json.dumps({'name': self.name, 'config': self.config, 'cls': type(self).__name__,
                                                    'n_users': self._n_users, 'n_items': self._n_items}).encode(),
                                        dtype=np.uint8)}
        if self.ftime in self.inter_feat:
            arrays['time'] = self.inter_feat[self.ftime].numpy()
        np.savez(path, **arrays)

    @classmethod
    def load_cache(cls, path):
        import json
        z = np.load(path if str(path).endswith('.npz') else str(path) + '.npz', allow_pickle=False)
        meta = json.loads(bytes(z['meta']).decode())
        self = cls.__new__(cls)
        self.name, self.config = meta['name'], meta['config']
        self.fuid = self.config['user_id_field'].split(':')[0]
        self.fiid = self.config['item_id_field'].split(':')[0]
        self.frating = self.config['rating_field'].split(':')[0]
        self.ftime = self.config['time_field'].split(':')[0] if self.config['time_field'] else None
        self._n_users, self._n_items = meta['n_users'], meta['n_items']
        self.field2tokens = {self.fuid: z['user_tokens'], self.fiid: z['item_tokens']}
        self.inter_feat = {self.fuid: torch.from_numpy(z['user']), self.fiid: torch.from_numpy(z['item']),
                           self.frating: torch.from_numpy(z['rating'])}
        if 'time' in z.files:
            self.inter_feat[self.ftime] = torch.from_numpy(z['time'])
        self._use_field = {self.fuid, self.fiid, self.frating}
        self.eval_mode = False
        return self

    def _read_atomic_files(self):
        import pandas as pd
        cfg = self.config
        if not cfg['data_dir'] or not cfg['inter_feat_name']:
            raise ValueError("config needs 'data_dir' and 'inter_feat_name' (RecStudio atomic-file layout)")
        sep = cfg['field_separator']
        names = [f.split(':')[0] for f in cfg.get('inter_feat_field', [cfg['user_id_field'], cfg['item_id_field'],
                                                                       cfg['rating_field'], cfg['time_field']]) if f]
        df = pd.read_csv(os.path.join(cfg['data_dir'], cfg['inter_feat_name']), sep=sep, header=cfg['inter_feat_header'],
                         names=names, dtype={self.fuid: str, self.fiid: str}, engine='python', index_col=False)
        df = df.dropna(how='any')
        extra = None
        if cfg['user_feat_name']:
            uf = cfg['user_feat_name'][0] if isinstance(cfg['user_feat_name'], list) else cfg['user_feat_name']
            udf = pd.read_csv(os.path.join(cfg['data_dir'], uf), sep=sep, header=cfg['user_feat_header'], dtype=str,
                              engine='python', index_col=False)
            extra = udf.iloc[:, 0].values.astype(str)
        times = df[self.ftime].values.astype(np.float64) if self.ftime in df else None
        ratings = df[self.frating].values.astype(np.float64) if self.frating in df else np.ones(len(df))
        return df[self.fuid].values.astype(str), df[self.fiid].values.astype(str), ratings, times, extra

    def _preprocess(self, users, items, ratings, times, extra_users):
        cfg = self.config
        keep = np.ones(len(users), dtype=bool)
        if cfg['low_rating_thres'] is not None:                       # _filter_ratings
            keep &= ratings >= cfg['low_rating_thres']
        users, items, ratings = users[keep], items[keep], ratings[keep]
        times = None if times is None else np.asarray(times)[keep]
        ucode, _ = _factorize(users)
        icode, _ = _factorize(items)
        if self.drop_dup:                                            # _drop_duplicated_pairs (keep first)
            pair = ucode.astype(np.int64) * (icode.max() + 1 if len(icode) else 1) + icode
            _, first = np.unique(pair, return_index=True)
            sel = np.sort(first)
            users, items, ratings = users[sel], items[sel], ratings[sel]
            times = None if times is None else times[sel]
            ucode, icode = ucode[sel], icode[sel]
        mu, mi = cfg['min_user_inter'], cfg['min_item_inter']
        if (mu and mu > 0) or (mi and mi > 0):                       # k-core, _filter :528-573
            keep = np.ones(len(users), dtype=bool)
            while True:
                ic = np.bincount(icode[keep], minlength=icode.max() + 1)
                k2 = keep & (ic[icode] >= (mi or 0))
                uc = np.bincount(ucode[k2], minlength=ucode.max() + 1)
                k3 = k2 & (uc[ucode] >= (mu or 0))
                if k3.sum() == keep.sum():
                    break
                keep = k3
            users, items, ratings = users[keep], items[keep], ratings[keep]
            times = None if times is None else times[keep]
        # token -> index by first appearance, [PAD] = 0 (_map_all_ids); the user file extends the id space
        utok = users if extra_users is None else np.concatenate([users.astype(str), np.asarray(extra_users, dtype=str)])
        ucode, utokens = _factorize(utok)
        icode, itokens = _factorize(items)
        self.field2tokens = {self.fuid: np.concatenate([['[PAD]'], utokens.astype(str)]),
                             self.fiid: np.concatenate([['[PAD]'], itokens.astype(str)])}
        self._n_users, self._n_items = len(utokens) + 1, len(itokens) + 1
        self.inter_feat = {self.fuid: torch.from_numpy(ucode[:len(users)].astype(np.int64) + 1),
                           self.fiid: torch.from_numpy(icode.astype(np.int64) + 1),
                           self.frating: torch.from_numpy(np.asarray(ratings, dtype=np.float32))}
        if times is not None:
            self.inter_feat[self.ftime] = torch.from_numpy(np.asarray(times, dtype=np.float64))

    # ------------------------------------------------------------------ properties
    @property
    def drop_dup(self):
        return self.config.get('drop_dup', True)

    @property
    def use_field(self):
        return self._use_field

    @use_field.setter
    def use_field(self, fields):
        self._use_field = set(fields)

    def drop_feat(self, keep_fields):
        if keep_fields:
            keep = set(keep_fields) | {self.frating}
            for k in list(self.inter_feat):
                if k not in keep:
                    del self.inter_feat[k]

    @property
    def num_users(self):
        return self._n_users

    @property
    def num_items(self):
        return self._n_items

    @property
    def num_inters(self):
        return len(self.inter_feat[self.fuid])

    def __len__(self):
        return len(self.data_index)

    @property
    def sample_length(self):
        if self.data_index.dim() > 1:
            return self.data_index[:, 2] - self.data_index[:, 1]
        raise ValueError('can not compute sample length for this dataset')

    @property
    def inter_feat_subset(self):
        if self.data_index.dim() > 1:
            return torch.cat([torch.arange(s, e) for s, e in zip(self.data_index[:, 1], self.data_index[:, 2])])
        return self.data_index

    @property
    def item_freq(self):
        """recstudio/data/dataset.py:1216-1230."""
        ids = self.inter_feat[self.fiid][self.inter_feat_subset]
        return torch.bincount(ids, minlength=self.num_items)

    def get_hist(self, isUser=True):
        """recstudio/data/dataset.py:1165-1189: padded interaction history + lengths."""
        u = self.inter_feat[self.fuid][self.inter_feat_subset]
        i = self.inter_feat[self.fiid][self.inter_feat_subset]
        key, val, n = (u, i, self.num_users) if isUser else (i, u, self.num_items)
        order = torch.sort(key, stable=True).indices
        count = torch.bincount(key, minlength=n)
        hist = torch.zeros(n, int(count.max()) if len(key) else 0, dtype=torch.int64)
        if len(key):
            sk = key[order]
            start = torch.cumsum(count, 0) - count
            col = torch.arange(len(sk)) - start[sk]
            hist[sk, col] = val[order]
        return hist, count

    # ------------------------------------------------------------------ build / split
    def build(self, binarized_rating_thres=None, fmeval=False, neg_count=None, sampler=None, shuffle=True,
              split_mode='user_entry', split_ratio=(0.8, 0.1, 0.1), **kwargs):
        if split_mode != 'user_entry':
            raise NotImplementedError("only split_mode='user_entry' (the reference default) is implemented")
        return self._build(list(split_ratio) if not isinstance(split_ratio, int) else split_ratio, shuffle, False)

    def _sort_and_shuffle(self, shuffle, rep):
        u = self.inter_feat[self.fuid].numpy()
        if self.drop_dup and not rep:
            pair = u * self.num_items + self.inter_feat[self.fiid].numpy()
            _, first = np.unique(pair, return_index=True)
            sel = torch.from_numpy(np.sort(first))
            self.inter_feat = {k: v[sel] for k, v in self.inter_feat.items()}
            u = self.inter_feat[self.fuid].numpy()
        if self.ftime is not None and self.ftime in self.inter_feat:
            order = np.lexsort((self.inter_feat[self.ftime].numpy(), u))     # stable, (user, time)
        else:
            order = np.argsort(u, kind='stable')
        u = u[order]
        uids, counts = np.unique(u, return_counts=True)
        if shuffle:                                                          # dataset.py:1001-1005
            starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
            order = order[np.concatenate([np.random.permutation(c) + s for s, c in zip(starts, counts)])]
        order = torch.from_numpy(order)
        self.inter_feat = {k: v[order] for k, v in self.inter_feat.items()}
        return uids, counts

    @staticmethod
    def _split_by_ratio(ratio, counts):
        """recstudio/data/dataset.py:728-752 (user_entry mode)."""
        splits = np.outer(counts, ratio).astype(np.int32)
        splits[:, 0] = counts - splits[:, 1:].sum(axis=1)
        for i in range(1, len(ratio)):
            idx = (splits[:, -i] == 0) & (splits[:, 0] > 1)
            splits[idx, -i] += 1
            splits[idx, 0] -= 1
        splits = np.hstack([np.zeros((len(counts), 1), dtype=np.int32), np.cumsum(splits, axis=1)])
        return np.concatenate([[0], np.cumsum(counts)[:-1]]).reshape(-1, 1) + splits

    @staticmethod
    def _split_leave_out(num, counts):
        """recstudio/data/dataset.py:769-802 (rep=True)."""
        m = len(counts)
        splits = np.ones((m, num + 1), dtype=np.int64)
        splits[:, 0] = counts - num
        for k in range(num):
            idx = splits[:, 0] < 1
            splits[idx, 0] += 1
            splits[idx, k] -= 1
        splits = np.hstack([np.zeros((m, 1), dtype=np.int64), np.cumsum(splits, axis=1)])
        return np.concatenate([[0], np.cumsum(counts)[:-1]]).reshape(-1, 1) + splits

    def _get_data_idx(self, splits, uids):
        """recstudio/data/dataset.py:804-815: flat row ids for train, [user, start, end] for val/test."""
        out = [torch.from_numpy(np.concatenate([np.arange(s, e) for s, e in zip(splits[:, 0], splits[:, 1])]))]
        for i in range(2, splits.shape[1]):
            rows = [[u, s, e] for u, s, e in zip(uids, splits[:, i - 1], splits[:, i]) if e > s]
            out.append(torch.tensor(rows, dtype=torch.int64).view(-1, 3))
        return out

    def _copy(self, idx):
        d = copy.copy(self)
        d.data_index = idx
        return d

    def _build(self, ratio_or_num, shuffle, rep):
        uids, counts = self._sort_and_shuffle(shuffle, rep)
        if isinstance(ratio_or_num, int):
            splits = self._split_leave_out(ratio_or_num, counts)
        else:
            splits = self._split_by_ratio(ratio_or_num, counts)
        datasets = [self._copy(i) for i in self._get_data_idx(splits, uids)]
        user_hist, user_count = datasets[0].get_hist(True)
        for d in datasets[:2]:
            d.user_hist, d.user_count = user_hist, user_count
        if len(datasets) > 2:                                                # dataset.py:1063-1067
            uh, uc = datasets[1].get_hist(True)
            datasets[-1].user_hist = torch.cat((user_hist, uh), dim=-1).sort(dim=-1, descending=True).values
            datasets[-1].user_count = uc + user_count
        return datasets

    # ------------------------------------------------------------------ batches
    def _get_pos_data(self, index):
        if self.data_index.dim() > 1:                                        # eval rows [user, start, end]
            idx = self.data_index[index]
            data = {self.fuid: idx[:, 0]}
            lens = (idx[:, 2] - idx[:, 1]).tolist()
            rows = torch.cat([torch.arange(s, e) for s, e in zip(idx[:, 1].tolist(), idx[:, 2].tolist())])
            data[self.fiid] = pad_sequence(self.inter_feat[self.fiid][rows].split(lens), batch_first=True)
            data[self.frating] = pad_sequence(self.inter_feat[self.frating][rows].split(lens), batch_first=True)
        else:
            idx = self.data_index[index]
            data = {k: v[idx] for k, v in self.inter_feat.items() if k in self._use_field}
        return data

    def __getitem__(self, index):
        data = self._get_pos_data(index)
        if self.eval_mode and 'user_hist' not in data:                       # dataset.py:903-906
            user_count = int(self.user_count[data[self.fuid]].max())
            data['user_hist'] = self.user_hist[data[self.fuid]][:, 0:user_count]
        return data

    def loader(self, batch_size, shuffle=True, num_workers=0, drop_last=False, ddp=False, device=None, rank=None, world=None):
        """``ddp=True`` (recstudio/data/dataset.py:1113-1114): ``batch_size`` is per rank; every rank draws the same GLOBAL
        index batches of ``batch_size * world`` samples and keeps its contiguous 1/world of each (``RankSlice``; rank and
        world size from the arguments, torch.distributed or the launcher's environment)."""
        r, w = _rank_world(rank, world) if ddp else (0, 1)
        if self.data_index.dim() > 1:
            sampler = SortedDataSampler(self, batch_size * w, shuffle, drop_last)
        else:
            sampler = DataSampler(self, batch_size * w, shuffle, drop_last)
        if ddp:
            sampler = RankSlice(sampler, r, w)
        return _Loader(self, sampler, device)

    def train_loader(self, batch_size, shuffle=True, num_workers=0, drop_last=False, ddp=False, device=None, rank=None,
                     world=None):
        self.eval_mode = False
        return self.loader(batch_size, shuffle, num_workers, drop_last, ddp, device, rank, world)

    def eval_loader(self, batch_size, num_workers=0, ddp=False, device=None, rank=None, world=None):
        """``ddp=True`` (dataset.py:1141-1142): this rank's contiguous part of every (length-sorted) global evaluation
        batch of ``batch_size * world`` users; ``batch['_n_valid']`` rows of it are real."""
        self.eval_mode = True
        r, w = _rank_world(rank, world) if ddp else (0, 1)
        sampler = SortedDataSampler(self, batch_size * w)
        if ddp:
            sampler = RankSlice(sampler, r, w)
        return _Loader(self, sampler, device)

    def device_train_loader(self, batch_size, shuffle=True, drop_last=False, device='cuda', ddp=False, rank=None, world=None):
        """Training loader whose data never leaves the GPU (see _DeviceLoader); ``ddp`` as in ``loader``."""
        if self.data_index.dim() > 1 and not isinstance(self, SeqDataset):
            raise ValueError('device_train_loader is for training splits')
        self.eval_mode = False
        r, w = _rank_world(rank, world) if ddp else (0, 1)
        return _DeviceLoader(self, batch_size, shuffle, drop_last, device, r, w)


class SeqDataset(TripletDataset):
    """recstudio/data/dataset.py:1369-1445: every prefix of a user's (time-ordered) sequence is a
    sample ``[user, start, end)`` -> target = item at ``end``; the window keeps the last
    ``max_seq_len`` items."""

    @property
    def drop_dup(self):
        return False

    def build(self, binarized_rating_thres=None, fmeval=False, neg_count=None, sampler=None, shuffle=True,
              split_mode='user_entry', split_ratio=2, test_rep=True, train_rep=True, **kwargs):
        if split_mode != 'user_entry' or not isinstance(split_ratio, int) or not (test_rep and train_rep):
            raise NotImplementedError('SeqDataset: only leave-n-out, user_entry, repeated items allowed')
        return self._build(split_ratio, False, True)

    def _get_data_idx(self, splits, uids):
        maxlen = self.config['max_seq_len'] or int((splits[:, -1] - splits[:, 0]).max())
        parts = [[] for _ in range(splits.shape[1] - 1)]
        for sp, u in zip(splits, uids):
            rows = np.array([[u, max(sp[0], i - maxlen), i] for i in range(sp[0], sp[-1])], dtype=np.int64)
            rel = sp - sp[0]
            for k, chunk in enumerate(np.split(rows[1:], rel[1:-1] - 1)):
                parts[k].append(chunk.reshape(-1, 3))
        return [torch.from_numpy(np.concatenate(p)) for p in parts]

    @property
    def inter_feat_subset(self):
        first = self.data_index[self.data_index[:, 2] - self.data_index[:, 1] == 1][:, 1]
        return torch.cat([first, self.data_index[:, 2]], dim=0)

    def _get_pos_data(self, index):
        idx = self.data_index[index]
        data = {self.fuid: idx[:, 0]}
        start, end = idx[:, 1], idx[:, 2]
        lens = (end - start).tolist()
        data['seqlen'] = end - start
        rows = torch.cat([torch.arange(s, e) for s, e in zip(start.tolist(), end.tolist())])
        for k in (self.fiid, self.frating):
            data['in_' + k] = pad_sequence(self.inter_feat[k][rows].split(lens), batch_first=True)
            data[k] = self.inter_feat[k][end]
        return data

    def segments(self, index=None):
        """(flat item column, start, end) -- the CSR view ``rsa_seg_gather`` consumes directly."""
        idx = self.data_index if index is None else self.data_index[index]
        return self.inter_feat[self.fiid], idx[:, 1].contiguous(), idx[:, 2].contiguous()


def synthetic_interactions(n_users, n_items, n_inter, alpha=1.0, seed=1):
    """Seeded synthetic interaction stream: users uniform, items Zipf(alpha) over a permuted id space
    (SURVEY.md section 8d).  Returns (user_ids, item_ids) int64 arrays with ids starting at 1."""
    rng = np.random.default_rng(seed)
    users = rng.integers(1, n_users + 1, size=n_inter)
    p = 1.0 / np.arange(1, n_items + 1) ** alpha
    p /= p.sum()
    items = rng.permutation(n_items)[rng.choice(n_items, size=n_inter, p=p)] + 1
    return users.astype(np.int64), items.astype(np.int64)
