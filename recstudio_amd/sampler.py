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

__all__ = ['Sampler', 'UniformSampler', 'MaskedUniformSampler', 'PopularSamplerModel', 'build_guide_table']


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


class PopularSamplerModel(Sampler):
    """recstudio/ann/sampler.py:224-258.  The fp32 tables are built with the very same torch
    CPU ops as the reference's constructor (so they are bit-identical to its registered
    buffers) and uploaded with the module; a guide table accelerates the inverse-CDF search."""

    def __init__(self, pop_count, scorer=None, mode=0, guide_log2=None):
        super().__init__(pop_count.shape[0], scorer)
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
            guide, self.guide_log2 = build_guide_table(self.table, guide_log2)
            self.register_buffer('guide', guide)
            self._register_pairs()

    def _register_pairs(self):
        # interleaved {table[i], pop_prob[i]} copy for the fused kernel (not part of the reference's state)
        self.register_buffer('table_prob', torch.stack([self.table, self.pop_prob], 1).contiguous(), persistent=False)
        # direct-lookup table, one self-contained 16-B entry per guide bucket (layout: rsa_common.hpp,
        # cdf_lookup_lut): {lo | search bit, table[lo] or +inf, pop_prob[lo], pop_prob[lo+1]} -- one HBM line
        # per sampled id instead of three dependent round trips (134 MB at 2^23 buckets)
        last = self.table.numel() - 1
        lo = self.guide.to(torch.int64)
        hi = torch.cat([lo[1:], lo[-1:]])
        span = hi - lo
        g0, g1 = lo.clamp(max=last), (lo + 1).clamp(max=last)
        x = (lo - (span >= 2).to(torch.int64) * (1 << 31)).to(torch.int32)
        y = torch.where(span == 0, torch.full_like(self.table[g0], float('inf')), self.table[g0])
        lut = torch.stack([x.view(torch.float32), y, self.pop_prob[g0], self.pop_prob[g1]], 1).contiguous()
        self.register_buffer('cdf_lut', lut, persistent=False)

    @classmethod
    def from_tables(cls, pop_prob, table, guide_log2=None):
        """Build from already-computed reference buffers (e.g. a loaded RecStudio checkpoint's
        ``sampler.pop_prob`` / ``sampler.table``) instead of recomputing them from counts."""
        self = cls.__new__(cls)
        Sampler.__init__(self, table.numel(), None)
        self.register_buffer('pop_prob', pop_prob.detach().clone().to(torch.float32))
        self.register_buffer('table', table.detach().clone().to(torch.float32))
        guide, self.guide_log2 = build_guide_table(self.table, guide_log2)
        self.register_buffer('guide', guide.to(self.table.device))
        self._register_pairs()
        return self

    def forward(self, query, num_neg, pos_items=None):
        with torch.no_grad():
            shape = tuple(query.shape[:-1])
            nq = int(np.prod(shape))
            neg, neg_prob = ops.sample_popular(self.table, self.pop_prob, self.guide, self.guide_log2, nq * num_neg,
                                               cdf_lut=getattr(self, 'cdf_lut', None))
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
