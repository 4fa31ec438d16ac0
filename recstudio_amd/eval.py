"""Ranking metrics on the bool hit matrix -- mirror of ``recstudio.eval`` for the six rank metrics
(recstudio/eval/__init__.py:9-165).  ``pred`` is bool [B, topk] ("the j-th ranked item is a hit"),
``target`` the ratings [B, T] of the ground-truth items (> 0 = relevant).  Tiny tensors; plain torch."""
import sys

import torch

__all__ = ['recall', 'precision', 'map', 'ndcg', 'mrr', 'hits', 'get_rank_metrics', 'metric_dict']


def recall(pred, target, k):
    count = (target > 0).sum(-1)
    return (pred[:, :k].sum(dim=-1).float() / count).mean()


def precision(pred, target, k):
    return (pred[:, :k].sum(dim=-1).float() / k).mean()


def map(pred, target, k):
    count = (target > 0).sum(-1)
    p = pred[:, :k].float()
    prec_at = p.cumsum(dim=-1) / torch.arange(1, k + 1, device=p.device).type_as(p)
    return ((prec_at * p).sum(dim=-1) / torch.minimum(count, k * torch.ones_like(count))).mean()


def _dcg(p, k):
    k = min(k, p.size(1))
    denom = torch.log2(torch.arange(k, device=p.device).type_as(p) + 2.0).view(1, -1)
    return (p[:, :k] / denom).sum(dim=-1)


def ndcg(pred, target, k):
    pred_dcg = _dcg(pred.float(), k)
    ideal = _dcg(torch.sort((target > 0).float(), descending=True)[0], k)
    irrelevant = torch.all(target <= sys.float_info.epsilon, dim=-1)
    # (same values as the reference's two masked assignments, eval/__init__.py:129-131, without their boolean-mask
    # indexing: each of those is a nonzero() -> a host synchronisation per metric per evaluation batch)
    return torch.where(irrelevant, torch.zeros_like(pred_dcg), pred_dcg / ideal).mean()


def mrr(pred, target, k):
    hit = pred[:, :k]
    first = hit.float().argmax(dim=-1) + 1
    out = torch.where(hit.any(dim=-1), 1.0 / first.float(), torch.zeros_like(first, dtype=torch.float))
    return out.mean()


def hits(pred, target, k):
    return torch.any(pred[:, :k] > 0, dim=-1).float().mean()


metric_dict = {'ndcg': ndcg, 'precision': precision, 'recall': recall, 'map': map, 'hit': hits, 'mrr': mrr}


def get_rank_metrics(metric):
    if not isinstance(metric, list):
        metric = [metric]
    return [(m, metric_dict[m]) for m in metric if m in metric_dict]
