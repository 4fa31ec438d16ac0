"""Negative samplers -- host-side mirror of ``recstudio.ann.sampler`` for the graded samplers.

Same class names, constructor arguments, ``forward`` signature and return
conventions as recstudio/ann/sampler.py:48-58 (Sampler), :81-114
(UniformSampler) and :224-258 (PopularSamplerModel); the sampling itself runs in
HIP (``rsa_sample_uniform`` / ``rsa_sample_popular``) on the torch device Philox
stream, so the ids equal what the reference's ``torch.randint`` / ``torch.rand``
+ ``torch.searchsorted`` produce on the same device for the same seed.
"""
from typing import Optional, Union

import numpy as np
import torch
from torch import Tensor

from . import ops

__all__ = ['Sampler', 'UniformSampler', 'MaskedUniformSampler', 'PopularSamplerModel', 'build_guide_table',
           'build_cdf_lines']


class Sampler(torch.nn.Module):
    """recstudio/ann/sampler.py:48-58."""

    def __init__(self, num_items, scorer_fn=None):
        super().__init__()
        self.num_items = num_items - 1   # remove padding (sampler.py:51)
        self.scorer = scorer_fn

    def update(self, item_embs, max_iter=30):
        pass

    def compute_item_p(self, query, pos_items):
        pass


def _query_shape(query, pos_items, device):
    if isinstance(query, int):                      # sampler.py:91-94
        dev = pos_items.device if pos_items is not None else device
        return (query,), dev
    if isinstance(query, Tensor):                   # sampler.py:95-99
        return tuple(query.shape[:-1]), query.device
    raise TypeError('`query` must be an int or a Tensor')


class UniformSampler(Sampler):
    """recstudio/ann/sampler.py:81-114: ids uniform in [1, num_items], int64 zero "log-probs"."""

    def forward(self, query: Union[Tensor, int], num_neg: int, pos_items: Optional[Tensor] = None,
                device: Optional[torch.device] = None):
        shape, dev = _query_shape(query, pos_items, device)
        nq = int(np.prod(shape))
        with torch.no_grad():
            neg = ops.sample_uniform(nq * num_neg, 1, self.num_items + 1, dev)
            neg = neg.view(*shape, num_neg)
            neg_prob = self.compute_item_p(None, neg)
            if pos_items is not None:
                return self.compute_item_p(None, pos_items), neg, neg_prob
            return neg, neg_prob

    def compute_item_p(self, query, pos_items):
        return ops.zero_logp_like(pos_items)        # sampler.py:113-114


class MaskedUniformSampler(Sampler):
    """recstudio/ann/sampler.py:187-214: uniform negatives excluding each user's history (rejection-free).
    BaseRetriever passes ``user_hist`` because the parameter is named in ``forward`` (baseretriever.py:225-228)."""

    def forward(self, query, num_neg, pos_items=None, user_hist=None):
        with torch.no_grad():
            if query.dim() == 2:
                neg = ops.sample_masked_uniform(user_hist, self.num_items, num_neg)
            elif query.dim() == 3:
                nq = query.size(1)
                neg = ops.sample_masked_uniform(user_hist, self.num_items, nq * num_neg).view(query.size(0), nq, num_neg)
            else:
                raise ValueError('`query` need to be 2-dimensional or 3-dimensional.')
            neg_prob = self.compute_item_p(query, neg)
            if pos_items is not None:
                return self.compute_item_p(query, pos_items), neg, neg_prob
            return neg, neg_prob

    def compute_item_p(self, query, pos_items):
        return ops.zero_logp_like(pos_items)            # -log(1), int64 like the reference (sampler.py:213-214)


def build_guide_table(table: Tensor, guide_log2: Optional[int] = None):
    """Cut-point ("guide") table for the fp32 CDF ``table``: guide[j] = first index i with
    table[i] >= j / K, K = 2**guide_log2, guide[K] = N.  Searching only inside
    [guide[b], guide[b+1]] for b = floor(u*K) returns exactly torch.searchsorted(table, u)
    (sampler.py:247): u*K is exact in fp32 (power-of-two scale) and the cut points j/K are compared with the
    fp32 CDF in float64, where both are exact."""
    n = table.numel()
    if guide_log2 is None:
        # ~1 item per bucket; LUT = 16 B per bucket: 134 MB at N = 1e7, 537 MB (the cap) at N = 1e8 -- measured best
        # of 2^23..2^26 there; 2^24 at N = 1e7 is slower than 2^23 (fewer searches but a wider random footprint)
        guide_log2 = int(min(25, max(4, int(np.ceil(np.log2(max(n, 2)))) - 1)))
    K = 1 << guide_log2
    cuts = torch.arange(K + 1, dtype=torch.float64) / K
    guide = torch.searchsorted(table.detach().cpu().to(torch.float64).contiguous(), cuts)
    guide[K] = n
    return guide.to(torch.int32), guide_log2


LINE_SLOTS = 12     # entry slots of one bucket line (rsa_common.hpp)


def build_cdf_lines(table: Tensor, pop_prob: Tensor, lines_log2: Optional[int] = None, max_bytes: int = 2 << 30,
                    overflow_target: float = 1e-2):
    """BUCKET LINES: the one-HBM-line form of ``torch.searchsorted(table, u)`` (layout: include/recstudio_amd.h,
    ``rsa_fused_args.cdf_lines``).  Bucket b covers u in [b, b+1) / 2**lines_log2; its 128-byte line lists the bucket's
    first distinct CDF values, then -- when there is room -- the first distinct entry ABOVE the bucket, each with its
    probability and its id as a 16-bit offset from the line's base id: 12 slots of {cdf fp32, prob fp32, delta u16} +
    base + count = 128 bytes.  The slot of a draw is #{cdf[i] < u}.  Items whose CDF value equals their predecessor's
    (zero probability, or a probability the fp32 cumsum absorbed) can never be returned by a lower-bound search and are
    left out, so a catalog where most items were never seen does not crowd the lines.  Cut points are compared in
    float64, where the fp32 CDF and b / 2**g are both exact -- same argument as ``build_guide_table``.

    ``lines_log2`` None: the smallest size (>= 2**4 buckets) for which at most ``overflow_target`` of the buckets
    (= of the draws, u being uniform) hold more than 11 distinct values (a 12th leaves no room for the entry above the
    bucket; a draw beyond the 12th falls back to a binary search), capped at ``max_bytes``.
    Returns (lines [2**g + 1, 32] float32 -- the last line is a sentinel closing the fallback ranges --, g)."""
    t = table.detach().cpu().to(torch.float32).contiguous()
    pp = pop_prob.detach().cpu().to(torch.float32).contiguous()
    n = t.numel()
    distinct = torch.ones(n, dtype=torch.bool)
    distinct[1:] = t[1:] > t[:-1]
    ids = torch.nonzero(distinct).flatten()                    # first index of every run of equal CDF values
    cdf = t[ids]
    nd = ids.numel()
    cdf64 = cdf.to(torch.float64)

    def cuts_of(g):
        K = 1 << g
        gd = torch.searchsorted(cdf64, torch.arange(K + 1, dtype=torch.float64) / K)     # first entry >= b / K
        gd[K] = nd
        return gd

    def crowded(g):
        c = cuts_of(g)
        return float(((c[1:] - c[:-1]) > LINE_SLOTS - 1).double().mean())

    if lines_log2 is None:
        g_max = max(4, int(np.floor(np.log2(max_bytes / 128))))
        g = int(min(g_max, max(4, int(np.ceil(np.log2(max(nd, 2) / 4.0))))))            # ~4 entries per bucket
        while g > 4 and crowded(g - 1) <= overflow_target:       # as small as the overflow target allows
            g -= 1
        while g < g_max and crowded(g) > overflow_target:
            g += 1
        lines_log2 = g
    gd = cuts_of(lines_log2)
    K = 1 << lines_log2
    lo, hi = gd[:-1], gd[1:]
    cnt = hi - lo
    last = n - 1
    as_f = lambda x: x.to(torch.int32).view(torch.float32)       # noqa: E731  int32 bit patterns in a float tensor
    lines = torch.zeros(K + 1, 32, dtype=torch.float32)
    slot_id = torch.zeros(K, LINE_SLOTS, dtype=torch.int64)
    for k in range(LINE_SLOTS):
        e = lo + k
        used = e <= hi                                            # real entries (e < hi) and the one above the bucket (e == hi)
        past = e >= nd                                            # nothing above: searchsorted returns n -> clamped to n - 1
        ec = e.clamp(max=nd - 1)
        idk = torch.where(past, torch.full_like(e, last), ids[ec])
        slot_id[:, k] = torch.where(used, idk, torch.zeros_like(idk))
        lines[:K, k] = torch.where(used & ~past, cdf[ec], torch.full((K,), float('inf')))
        lines[:K, 12 + k] = torch.where(used, pp[idk], torch.zeros(K))
    base = slot_id[:, 0]
    used_cols = torch.arange(LINE_SLOTS).view(1, -1) <= cnt.view(-1, 1)
    delta = torch.where(used_cols, slot_id - base.view(-1, 1), torch.zeros_like(slot_id))
    wide = (delta > 0xffff).any(1)                                # ids too far apart for 16-bit offsets: searched instead
    delta = delta.clamp(min=0, max=0xffff)
    packed = (delta[:, 0::2] | (delta[:, 1::2] << 16))            # two uint16 per word, even slot in the low half
    packed = torch.where(packed >= (1 << 31), packed - (1 << 32), packed)
    lines[:K, 24:30] = packed.to(torch.int32).view(torch.float32)
    # fallback range in `table`: [first id at or above the bucket's lower cut, the next line's]
    first = torch.where(lo < nd, ids[lo.clamp(max=nd - 1)], torch.full_like(lo, last))
    lines[:K, 30] = as_f(torch.where(wide, first, base))
    lines[:K, 31] = as_f(torch.where(wide, torch.full_like(cnt, -1), cnt))
    lines[K, 30] = as_f(torch.tensor([last]))[0]
    lines[K, :12] = float('inf')
    return lines, lines_log2


def _rebuild_derived(module, incompatible_keys):
    """load_state_dict post-hook of PopularSamplerModel (a module-level function: a lambda in the hook dict would make the
    module unpicklable -- torch.save(model), mp.spawn arguments)."""
    module._register_pairs()


class PopularSamplerModel(Sampler):
    """recstudio/ann/sampler.py:224-258.  The fp32 tables are built with the very same torch
    CPU ops as the reference's constructor (so they are bit-identical to its registered
    buffers) and uploaded with the module; a guide table accelerates the inverse-CDF search."""

    # derived lookup structures: rebuilt from pop_prob / table, never part of the state dict (the reference's
    # checkpoint holds exactly `pop_prob` and `table`, sampler.py:239-241)
    LINES_MIN_ITEMS = 1 << 16        # below this the tables live in L2 and every form costs the same

    def __init__(self, pop_count, scorer=None, mode=0, guide_log2=None, lookup='auto', lines_log2=None):
        """``lookup``: 'lines' (bucket lines, one HBM line per draw), 'lut' (16-byte direct-lookup table + 4-wide
        probe), 'guide' (guide table + binary search) or 'auto' (lines for catalogs of LINES_MIN_ITEMS items or
        more).  All return torch.searchsorted's index."""
        super().__init__(pop_count.shape[0], scorer)
        self.lookup, self._guide_log2_arg, self._lines_log2_arg = lookup, guide_log2, lines_log2
        with torch.no_grad():
            pop_count = torch.as_tensor(pop_count).detach().to('cpu', torch.float)
            if mode == 0:
                pop_count = torch.log(pop_count + 1)
            elif mode == 1:
                pop_count = torch.log(pop_count + 1) + 1e-6
            elif mode == 2:
                pop_count = pop_count ** 0.75
            pop_count[0] = 1                                              # sampler.py:237
            self.register_buffer('pop_prob', pop_count / pop_count.sum())
            self.register_buffer('table', torch.cumsum(self.pop_prob, dim=0))
            self.pop_prob[-1] = 1.0                                       # sampler.py:241
            self._register_pairs()
        self.register_load_state_dict_post_hook(_rebuild_derived)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # checkpoints of an earlier revision carried the derived `guide` buffer: not part of the state any more
        state_dict.pop(prefix + 'guide', None)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _use_lines(self):
        if self.lookup == 'auto':
            return self.table.numel() >= self.LINES_MIN_ITEMS
        return self.lookup == 'lines'

    @torch.no_grad()
    def _register_pairs(self):
        """(Re)build every derived buffer from ``table`` / ``pop_prob`` -- also after ``load_state_dict`` replaced them,
        so that the kernels never sample from a stale structure while ``compute_item_p`` reads the new ``pop_prob``.
        None of them is persistent: the state dict is the reference's (``pop_prob``, ``table``)."""
        dev = self.table.device
        table, prob = self.table.detach().cpu(), self.pop_prob.detach().cpu()
        # interleaved {table[i], pop_prob[i]} copy (binary-search fallbacks share cache lines with the probability)
        self.register_buffer('table_prob', torch.stack([table, prob], 1).contiguous().to(dev), persistent=False)
        guide = lut = lines = None
        self.guide_log2 = self.lines_log2 = 0
        if self._use_lines():
            lines, self.lines_log2 = build_cdf_lines(table, prob, self._lines_log2_arg)
        else:
            guide, self.guide_log2 = build_guide_table(table, self._guide_log2_arg)
            if self.lookup in ('auto', 'lut'):
                # direct-lookup table, one self-contained 16-B entry per guide bucket (layout: rsa_common.hpp,
                # cdf_lookup_lut): {lo | search bit, table[lo] or +inf, pop_prob[lo], pop_prob[lo+1]}
                last = table.numel() - 1
                lo = guide.to(torch.int64)
                hi = torch.cat([lo[1:], lo[-1:]])
                span = hi - lo
                g0, g1 = lo.clamp(max=last), (lo + 1).clamp(max=last)
                x = (lo - (span >= 2).to(torch.int64) * (1 << 31)).to(torch.int32)
                y = torch.where(span == 0, torch.full_like(table[g0], float('inf')), table[g0])
                lut = torch.stack([x.view(torch.float32), y, prob[g0], prob[g1]], 1).contiguous()
        for name, t in (('guide', guide), ('cdf_lut', lut), ('cdf_lines', lines)):
            self.register_buffer(name, None if t is None else t.to(dev), persistent=False)

    @classmethod
    def from_tables(cls, pop_prob, table, guide_log2=None, lookup='auto', lines_log2=None):
        """Build from already-computed reference buffers (e.g. a loaded RecStudio checkpoint's
        ``sampler.pop_prob`` / ``sampler.table``) instead of recomputing them from counts."""
        self = cls.__new__(cls)
        Sampler.__init__(self, table.numel(), None)
        self.lookup, self._guide_log2_arg, self._lines_log2_arg = lookup, guide_log2, lines_log2
        self.register_buffer('pop_prob', pop_prob.detach().clone().to(torch.float32))
        self.register_buffer('table', table.detach().clone().to(torch.float32))
        self._register_pairs()
        self.register_load_state_dict_post_hook(_rebuild_derived)
        return self

    def lookup_kwargs(self):
        """The popularity tables as keyword arguments of ``ops.fused_forward`` / ``ops.sample_popular``."""
        return dict(table=self.table, pop_prob=self.pop_prob, guide=self.guide, guide_log2=self.guide_log2,
                    table_prob=getattr(self, 'table_prob', None), cdf_lut=getattr(self, 'cdf_lut', None),
                    cdf_lines=getattr(self, 'cdf_lines', None), lines_log2=getattr(self, 'lines_log2', 0))

    def forward(self, query, num_neg, pos_items=None):
        with torch.no_grad():
            shape = tuple(query.shape[:-1])
            nq = int(np.prod(shape))
            neg, neg_prob = ops.sample_popular(self.table, self.pop_prob, self.guide, self.guide_log2, nq * num_neg,
                                               cdf_lut=getattr(self, 'cdf_lut', None),
                                               cdf_lines=getattr(self, 'cdf_lines', None),
                                               lines_log2=getattr(self, 'lines_log2', 0))
            neg = neg.view(*shape, num_neg)
            neg_prob = neg_prob.view(*shape, num_neg)
            if pos_items is not None:
                return self.compute_item_p(query, pos_items), neg, neg_prob
            return neg, neg_prob

    def compute_item_p(self, query, pos_items):
        return ops.item_logp(self.pop_prob, pos_items)                    # sampler.py:257-258


class RetrieverSampler(Sampler):
    """recstudio/ann/sampler.py:61-78 (IRGAN): negatives drawn by ANOTHER retriever's ``sampling(method=...)`` --
    by default from its softmax over the whole catalog ('brute'), scored on the full-score kernels."""

    def __init__(self, num_items, retriever=None, method='brute', t=1):
        super().__init__(num_items)
        self.retriever, self.method, self.T = retriever, method, t

    def update(self, item_embs, max_iter=30):
        self.retriever._update_item_vector()

    def forward(self, batch, num_neg, pos_items=None, excluding_hist=False):
        (log_pos_prob, neg_id, log_neg_prob), _ = self.retriever.sampling(
            batch=batch, num_neg=num_neg, excluding_hist=excluding_hist, method=self.method, return_query=False, t=self.T)
        return log_pos_prob.detach(), neg_id.detach(), log_neg_prob.detach()
