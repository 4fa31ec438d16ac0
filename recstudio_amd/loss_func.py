"""Losses -- host-side mirror of ``recstudio.model.loss_func`` for BPR / SampledSoftmax / Softmax.

Class taxonomy and ``forward`` parameter NAMES follow the reference
(recstudio/model/loss_func.py:6-28, :39-47, :50-63, :80-90) because
``BaseRetriever.training_step`` calls ``loss_fn(**score)`` by keyword
(baseretriever.py:399-404).  Value and gradient come from ``rsa_pairwise_loss`` in one
pass; autograd sees a single node.
"""
import torch

from . import _native as nat
from . import ops

__all__ = ['FullScoreLoss', 'PairwiseLoss', 'PointwiseLoss', 'BPRLoss', 'SampledSoftmaxLoss', 'SoftmaxLoss',
           'BinaryCrossEntropyLoss']


class FullScoreLoss(torch.nn.Module):
    def forward(self, label, pos_score, all_score):
        pass


class PairwiseLoss(torch.nn.Module):
    def forward(self, label, pos_score, log_pos_prob, neg_score, log_neg_prob):
        pass


class PointwiseLoss(torch.nn.Module):
    def forward(self, label, pos_score):
        raise NotImplementedError(f'{type(self).__name__} is an abstract class')


class _PairwiseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, pos_score, neg_score, pos_logp, neg_logp):
        need = pos_score.requires_grad or neg_score.requires_grad
        loss, dpos, dneg, _ = ops.pairwise_loss(kind, pos_score, neg_score, pos_logp, neg_logp, want_grad=True)
        ctx.save_for_backward(dpos, dneg)
        return loss

    @staticmethod
    def backward(ctx, g):
        dpos, dneg = ctx.saved_tensors
        return None, dpos * g, dneg * g, None, None


def _as_f32_or_none(t):
    """The reference's UniformSampler hands int64 zeros as log-probs (sampler.py:113-114)."""
    if t is None:
        return None
    if not t.is_floating_point():
        # zeros created by this package are recognised by identity; anything else is cast (no host sync either way)
        return None if ops.is_known_zero(t) else t.to(torch.float32)
    return t


def _check_shapes(pos_score, neg_score):
    if pos_score.dim() != neg_score.dim() - 1 or tuple(pos_score.shape) != tuple(neg_score.shape[:-1]):
        raise NotImplementedError(
            'recstudio_amd pairwise losses need pos_score [..] with neg_score [.., n] (one negative set per '
            f'positive); got {tuple(pos_score.shape)} and {tuple(neg_score.shape)}')


class BPRLoss(PairwiseLoss):
    """recstudio/model/loss_func.py:50-59 (dns=False)."""

    def __init__(self, dns=False):
        super().__init__()
        if dns:
            raise NotImplementedError('BPRLoss(dns=True) is outside the hot path this package covers')
        self.dns = dns

    def forward(self, label, pos_score, log_pos_prob, neg_score, log_neg_prob):
        _check_shapes(pos_score, neg_score)
        return _PairwiseFn.apply(nat.LOSS_BPR, pos_score, neg_score, None, None)


class BinaryCrossEntropyLoss(PairwiseLoss):
    """recstudio/model/loss_func.py:100-132 (dns=False): SASRec's default loss (seq/sasrec.py:117-119)."""

    def __init__(self, dns=False):
        super().__init__()
        if dns:
            raise NotImplementedError('BinaryCrossEntropyLoss(dns=True) is outside the path this package covers')
        self.dns = dns

    def forward(self, label, pos_score, log_pos_prob, neg_score, log_neg_prob):
        _check_shapes(pos_score, neg_score)
        return _PairwiseFn.apply(nat.LOSS_BCE, pos_score, neg_score, None, None)


class _PairwiseExFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, pos_score, neg_score, pos_logp, neg_logp, p0, p1):
        loss, dpos, dneg = ops.pairwise_loss_ex(kind, pos_score, neg_score, pos_logp, neg_logp, p0, p1)
        ctx.save_for_backward(dpos, dneg)
        return loss

    @staticmethod
    def backward(ctx, g):
        dpos, dneg = ctx.saved_tensors
        return None, dpos * g, dneg * g, None, None, None, None


class WeightedBPRLoss(PairwiseLoss):
    """recstudio/model/loss_func.py:93-97: BPR with importance weights softmax(neg_score - log_neg_prob)."""

    def forward(self, label, pos_score, log_pos_prob, neg_score, log_neg_prob):
        _check_shapes(pos_score, neg_score)
        return _PairwiseExFn.apply(nat.LOSS_WBPR, pos_score, neg_score, None, _as_f32_or_none(log_neg_prob), 0.0, 0.0)


class WeightedBinaryCrossEntropyLoss(BinaryCrossEntropyLoss):
    """recstudio/model/loss_func.py:135-137."""

    def forward(self, label, pos_score, log_pos_prob, neg_score, log_neg_prob):
        _check_shapes(pos_score, neg_score)
        return _PairwiseExFn.apply(nat.LOSS_WBCE, pos_score, neg_score, None, _as_f32_or_none(log_neg_prob), 0.0, 0.0)


class HingeLoss(PairwiseLoss):
    """recstudio/model/loss_func.py:140-154 with num_items=None (the num_items branch of the reference takes the mean
    of a bool tensor, which torch rejects)."""

    def __init__(self, margin=2, num_items=None):
        super().__init__()
        if num_items is not None:
            raise NotImplementedError('HingeLoss(num_items=...) does not run in the reference either (mean of a bool tensor)')
        self.margin, self.n_items = margin, num_items

    def forward(self, label, pos_score, log_pos_prob, neg_score, log_neg_prob):
        _check_shapes(pos_score, neg_score)
        return _PairwiseExFn.apply(nat.LOSS_HINGE, pos_score, neg_score, None, None, float(self.margin), 0.0)


class NCELoss(PairwiseLoss):
    """recstudio/model/loss_func.py:163-168 (pos_score [B], neg_score [B, n])."""

    def forward(self, label, pos_score, log_pos_prob, neg_score, log_neg_prob):
        if pos_score.dim() != 1 or neg_score.dim() != 2:
            raise NotImplementedError('NCELoss: the reference sums the negatives over dim 1, i.e. expects [B] and [B, n]')
        _check_shapes(pos_score, neg_score)
        return _PairwiseExFn.apply(nat.LOSS_NCE, pos_score, neg_score, _as_f32_or_none(log_pos_prob),
                                   _as_f32_or_none(log_neg_prob), 0.0, 0.0)


class CCLLoss(PairwiseLoss):
    """recstudio/model/loss_func.py:171-186."""

    def __init__(self, margin=0.8, neg_weight=0.3):
        super().__init__()
        self.margin, self.neg_weight = margin, neg_weight

    def forward(self, label, pos_score, log_pos_prob, neg_score, log_neg_prob):
        _check_shapes(pos_score, neg_score)
        return _PairwiseExFn.apply(nat.LOSS_CCL, pos_score, neg_score, None, None, float(self.margin),
                                   float(self.neg_weight))


class _SharedSSMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos_score, neg_score, pos_logp, neg_logp):
        loss, dpos, dneg = ops.ssm_shared_loss(pos_score, neg_score, pos_logp, neg_logp)
        ctx.save_for_backward(dpos, dneg)
        return loss

    @staticmethod
    def backward(ctx, g):
        dpos, dneg = ctx.saved_tensors
        return dpos * g, dneg * g, None, None


class SampledSoftmaxLoss(PairwiseLoss):
    """recstudio/model/loss_func.py:80-90."""

    def forward(self, label, pos_score, log_pos_prob, neg_score, log_neg_prob):
        if pos_score.dim() == neg_score.dim() == 2:      # several positives sharing one negative set (:84-89)
            return _SharedSSMFn.apply(pos_score, neg_score, _as_f32_or_none(log_pos_prob),
                                      _as_f32_or_none(log_neg_prob))
        _check_shapes(pos_score, neg_score)
        return _PairwiseFn.apply(nat.LOSS_SSM, pos_score, neg_score, _as_f32_or_none(log_pos_prob),
                                 _as_f32_or_none(log_neg_prob))


class InfoNCELoss(SampledSoftmaxLoss):
    """recstudio/model/loss_func.py:157-160: sampled softmax with the proposal log-probabilities zeroed."""

    def forward(self, label, pos_score, log_pos_prob, neg_score, log_neg_prob):
        return super().forward(label, pos_score, None, neg_score, None)


class _MeanLseFn(torch.autograd.Function):
    """mean over rows of logsumexp(all_score[m, :]) with the softmax gradient produced in the same call."""

    @staticmethod
    def forward(ctx, all_score):
        x = all_score.reshape(-1, all_score.shape[-1])
        lse, sm = ops.row_lse(x, want_softmax=all_score.requires_grad, scale=1.0 / x.shape[0])
        ctx.shape = all_score.shape
        ctx.save_for_backward(sm)
        return lse.mean()

    @staticmethod
    def backward(ctx, g):
        (sm,) = ctx.saved_tensors
        return (sm * g).view(ctx.shape)


class _RowLseFn(torch.autograd.Function):
    """logsumexp(all_score[m, :]) per row (rsa_row_lse), softmax kept for the backward."""

    @staticmethod
    def forward(ctx, all_score):
        x = all_score.reshape(-1, all_score.shape[-1])
        lse, sm = ops.row_lse(x, want_softmax=all_score.requires_grad, scale=1.0)
        ctx.shape = all_score.shape
        ctx.save_for_backward(sm)
        return lse.view(all_score.shape[:-1])

    @staticmethod
    def backward(ctx, g):
        (sm,) = ctx.saved_tensors
        return (sm * g.reshape(-1, 1)).view(ctx.shape)


class SoftmaxLoss(FullScoreLoss):
    """recstudio/model/loss_func.py:39-47.  First branch (all_score one dim more than pos_score):
    mean(logsumexp(all_score, -1) - pos_score).  Second branch (same rank: several positives [B, L] per score row
    [B, N]): per row the mean over its non-padded positives of logsumexp - pos, padded (-inf) positives dropped."""

    def forward(self, label, pos_score, all_score):
        if all_score.dim() > pos_score.dim():
            return _MeanLseFn.apply(all_score) - pos_score.mean()
        lse = _RowLseFn.apply(all_score)                                     # the [B, N] reduction runs in HIP
        output = lse.unsqueeze(-1) - pos_score                               # [B, L]: a few elementwise torch ops
        notpadnum = torch.logical_not(torch.isinf(pos_score)).float().sum(-1)
        output = torch.nan_to_num(output, posinf=0).sum(-1) / notpadnum
        return torch.mean(output)
