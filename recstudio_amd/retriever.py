"""``BaseRetriever`` -- host-side mirror of recstudio/model/basemodel/baseretriever.py (the class the
reference's READMEs call TwoTowerRecommender / ItemTowerRecommender) with its forward dispatched to
the fused HIP path.

Kept from the reference: constructor kwargs (``item_encoder``, ``query_encoder``, ``scorer``,
``sampler``, ``loss``; baseretriever.py:14-43, recommender.py:48-56), the overridable hooks
(``_get_dataset_class``, ``_get_item_encoder``, ``_get_query_encoder``, ``_get_score_func``,
``_get_loss_func``, ``_get_sampler``, ``_set_data_field``), ``forward`` with its ``return_*`` flags
and output dict layout (:142-192), ``sampling`` / ``_sample`` for ``method='none'`` (:204-278),
``topk`` (:374-397), ``training_step`` / ``validation_step`` / ``test_step`` / ``_test_step``
(:399-431), ``fit`` / ``evaluate`` and the config keys read on the path.  The train/eval loop around
it is a minimal compatible one (the reference's Recommender loop is out of scope).

Dispatch rule (mirrors the reference's own check at baseretriever.py:122): the fused kernels are
used when the item tower is a plain ``nn.Embedding`` over the item id, the scorer is exactly
InnerProductScorer / CosineScorer, the sampler exactly UniformSampler / PopularSamplerModel (or None
for full-score losses) and ``sampling_method == 'none'``; anything else goes through the per-plugin
path (sampler / scorer / loss objects called one by one, like the reference does).
"""
import copy
import inspect
import logging
import os
import time
from typing import Dict

import numpy as np
import torch

from . import _native as nat
from . import eval as rs_eval
from . import ops
from .dataset import SeqDataset, TripletDataset
from .fused import fused_bpr_loss, fused_ssm_loss, retriever_scores
from .loss_func import (BinaryCrossEntropyLoss, BPRLoss, FullScoreLoss, PairwiseLoss, PointwiseLoss, SampledSoftmaxLoss,
                        SoftmaxLoss)
from .sampler import PopularSamplerModel, Sampler, UniformSampler
from .scorer import CosineScorer, EuclideanScorer, InnerProductScorer, full_lse

__all__ = ['BaseRetriever', 'TwoTowerRecommender', 'ItemTowerRecommender', 'BPR', 'SASRec', 'default_config',
           'seed_everything']


def default_config():
    """recstudio/model/basemodel/basemodel.yaml."""
    return {
        'data': {'binarized_rating_thres': None, 'fm_eval': False, 'neg_count': 0, 'sampler': None, 'shuffle': True,
                 'split_mode': 'user_entry', 'split_ratio': [0.8, 0.1, 0.1]},
        'model': {'embed_dim': 64, 'item_bias': False},
        'train': {'accelerator': 'gpu', 'ann': None, 'batch_size': 512, 'early_stop_mode': 'max',
                  'early_stop_patience': 10, 'epochs': 1000, 'gpu': 1, 'grad_clip_norm': None,
                  'init_method': 'xavier_normal', 'item_batch_size': 1024, 'learner': 'adam', 'learning_rate': 0.001,
                  'num_threads': 10, 'sampling_method': 'none', 'sampler': 'uniform', 'negative_count': 0,
                  'excluding_hist': False, 'scheduler': None, 'seed': 2022, 'weight_decay': 0.0,
                  'tensorboard_path': None, 'sparse_grad': False, 'device_loader': True, 'fused_optimizer': None,
                  'shard_slices': 1, 'shard_layout': 'block', 'shard_owner_loss': True, 'shard_init': 'auto',
                  'shard_lookahead': False, 'shard_deterministic': True, 'shard_rows_share': None,
                  'fused_prefetch': True},
        'eval': {'batch_size': 128, 'cutoff': [5, 10, 20], 'val_metrics': ['ndcg', 'recall'], 'val_n_epoch': 1,
                 'test_metrics': ['ndcg', 'recall', 'precision', 'map', 'mrr', 'hit'], 'topk': 100,
                 'save_path': './saved/'},
    }


def seed_everything(seed):
    """recstudio/utils/utils.py:334-381: python / numpy / torch CPU + all device generators."""
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    return seed


def _device_init_block(plan, rank, n_items, d, seed, method, device, rows_per_chunk=1 << 16):
    """This rank's rows of an [n_items, d] embedding table initialised as recstudio/model/init.py does (xavier_normal /
    xavier_uniform over the FULL shape, normal(0.02)) WITHOUT ever holding the full table: the table is defined chunk by
    chunk of ``rows_per_chunk`` GLOBAL rows, chunk c drawn from its own generator stream (seed, c) on the device, and a
    rank generates the chunks its rows fall into and keeps those rows.  The table is therefore a function of (seed, shape)
    alone: any world size, either row layout, gives the rows of the same table (at 1e8 x 128 the host copy the
    slice-after-init form needs is 51 GB per rank)."""
    n_local = plan.n_local(rank)
    out = torch.empty(n_local, d, dtype=torch.float32, device=device)
    gen = torch.Generator(device=device)
    if method == 'xavier_uniform':
        bound = (6.0 / (n_items + d)) ** 0.5
    std = (2.0 / (n_items + d)) ** 0.5 if method == 'xavier_normal' else 0.02
    R, G = int(rows_per_chunk), plan.world
    lo, hi = (0, n_items) if plan.interleaved else plan.bounds(rank)
    for c in range(lo // R, (max(hi, lo + 1) - 1) // R + 1):
        a, b = c * R, min((c + 1) * R, n_items)
        if b <= a:
            break
        gen.manual_seed((int(seed) * 1_000_003 + c) & 0x7fffffffffffffff)
        if method == 'xavier_uniform':
            vals = (torch.rand(R, d, generator=gen, device=device) * 2 - 1) * bound
        else:
            vals = torch.randn(R, d, generator=gen, device=device) * std
        if plan.interleaved:
            first = (rank - a) % G                     # first row of the chunk this rank owns
            rows = vals[first:b - a:G]
            at = (a + first - rank) // G
            out[at:at + rows.shape[0]] = rows
        else:
            u, v = max(a, lo), min(b, hi)
            if v > u:
                out[u - lo:v - lo] = vals[u - a:v - a]
    return out


def _init_weights(module, method):
    """recstudio/model/init.py: xavier_normal / xavier_uniform / normal(0.02); the padding row is zeroed."""
    if isinstance(module, torch.nn.Embedding) and module.weight.is_meta:
        return          # a sharded catalog: its rows are drawn on the owners' devices (_device_init_block)
    if isinstance(module, torch.nn.Embedding):
        if method == 'xavier_normal':
            torch.nn.init.xavier_normal_(module.weight.data)
        elif method == 'xavier_uniform':
            torch.nn.init.xavier_uniform_(module.weight.data)
        else:
            module.weight.data.normal_(mean=0.0, std=0.02)
        if module.padding_idx is not None:
            torch.nn.init.constant_(module.weight.data[module.padding_idx], 0.)
    elif isinstance(module, torch.nn.Linear):
        if method == 'xavier_normal':
            torch.nn.init.xavier_normal_(module.weight.data)
        elif method == 'xavier_uniform':
            torch.nn.init.xavier_uniform_(module.weight.data)
        else:
            module.weight.data.normal_(mean=0.0, std=0.02)
        if module.bias is not None:
            torch.nn.init.constant_(module.bias.data, 0)
    elif isinstance(module, torch.nn.LayerNorm):
        module.bias.data.zero_()
        module.weight.data.fill_(1.0)


class BaseRetriever(torch.nn.Module):
    def __init__(self, config: Dict = None, **kwargs):
        super().__init__()
        self.config = default_config()
        if config is not None:
            for group, values in config.items():
                if isinstance(values, dict) and group in self.config:
                    self.config[group].update(values)
                else:
                    self.config[group] = values
        if self.config['train']['seed'] is not None:
            seed_everything(self.config['train']['seed'])          # recommender.py:34-35
        self.embed_dim = self.config['model']['embed_dim']
        self.logged_metrics = {}
        self.logger = logging.getLogger('recstudio_amd')
        for key, attr in (('item_encoder', 'item_encoder'), ('query_encoder', 'query_encoder')):
            if key in kwargs:
                assert isinstance(kwargs[key], torch.nn.Module), f'{key} must be torch.nn.Module'
                setattr(self, attr, kwargs[key])
            else:
                setattr(self, attr, None)
        if 'scorer' in kwargs:
            assert isinstance(kwargs['scorer'], torch.nn.Module), 'scorer must be torch.nn.Module'
            self.score_func = kwargs['scorer']
        else:
            self.score_func = self._get_score_func()
        if 'sampler' in kwargs:
            assert isinstance(kwargs['sampler'], Sampler), 'sampler must be recstudio_amd.Sampler'
            self.sampler = kwargs['sampler']
        else:
            self.sampler = None
        if 'loss' in kwargs:
            assert isinstance(kwargs['loss'], (FullScoreLoss, PairwiseLoss, PointwiseLoss)), \
                'loss should be a FullScoreLoss, PairwiseLoss or PointwiseLoss'
            self.loss_fn = kwargs['loss']
        else:
            self.loss_fn = None
        if self.config['train']['ann'] is not None:
            raise NotImplementedError('ANN indexes (train.ann) are outside the path this package covers')
        self.use_index = False
        self.ckpt_path = None

    # ------------------------------------------------------------------ hooks (reference API)
    def _get_dataset_class():
        return TripletDataset

    def _set_data_field(self, data):
        data.use_field = set([data.fuid, data.fiid, data.frating])

    def _get_item_encoder(self, train_data):
        return torch.nn.Embedding(train_data.num_items, self.embed_dim, padding_idx=0)

    def _get_query_encoder(self, train_data):
        if self.fuid in self.query_fields:
            return torch.nn.Embedding(train_data.num_users, self.embed_dim, padding_idx=0)
        raise ValueError('query_encoder missing.')

    def _get_score_func(self):
        return InnerProductScorer()

    def _get_loss_func(self):
        return None

    def _get_sampler(self, train_data):
        return UniformSampler(train_data.num_items)

    # ------------------------------------------------------------------ model set-up
    def _init_model(self, train_data, drop_unused_field=True):
        self._set_data_field(train_data)
        self.fields = train_data.use_field
        self.frating = train_data.frating
        assert self.frating in self.fields, 'rating field is required.'
        if drop_unused_field:
            train_data.drop_feat(self.fields)
        self.fiid, self.fuid = train_data.fiid, train_data.fuid
        self.item_fields = {self.fiid} & set(self.fields)
        assert self.fiid in self.item_fields, 'item id is required to use.'
        self.query_fields = {self.fuid} & set(self.fields)
        if isinstance(train_data, SeqDataset):
            self.query_fields = self.query_fields | {'in_' + f for f in self.item_fields} | {'seqlen'}
        self.neg_count = self.config['train']['negative_count']
        if self.loss_fn is None:
            if 'train_data' in inspect.signature(self._get_loss_func).parameters:
                self.loss_fn = self._get_loss_func(train_data)
            else:
                self.loss_fn = self._get_loss_func()
        if not self.item_encoder and getattr(self, '_item_table_on_meta', False):
            # multi-GPU fit over a big catalog: the [N, d] table is never materialised on the host -- its shape only (a
            # meta tensor); every rank fills its own rows on its device (_setup_shard, _device_init_block)
            with torch.device('meta'):
                self.item_encoder = self._get_item_encoder(train_data)
        self.item_encoder = self._get_item_encoder(train_data) if not self.item_encoder else self.item_encoder
        self.query_encoder = self._get_query_encoder(train_data) if not self.query_encoder else self.query_encoder
        self.sampler = self._get_sampler(train_data) if not self.sampler else self.sampler

    def _init_parameter(self):
        method = self.config['train']['init_method']
        for _, module in self.named_children():
            module.apply(lambda m: _init_weights(m, method))

    def _get_item_feat(self, data):
        return data[self.fiid] if isinstance(data, dict) else data

    def _get_query_feat(self, data):
        if isinstance(data, dict):
            if len(self.query_fields) == 1:
                return data[list(self.query_fields)[0]]
            return {f: v for f, v in data.items() if f in self.query_fields}
        return data

    def _get_item_vector(self):
        if len(self.item_fields) == 1 and isinstance(self.item_encoder, torch.nn.Embedding):
            return self.item_encoder.weight[1:]                        # baseretriever.py:122-123
        raise NotImplementedError('item towers other than nn.Embedding over the item id are not covered')

    def _update_item_vector(self):
        item_vector = self._get_item_vector()
        if not hasattr(self, 'item_vector'):
            self.register_buffer('item_vector', item_vector.detach().clone())
        else:
            self.item_vector = item_vector

    # ------------------------------------------------------------------ forward
    def _fused_ok(self):
        return (isinstance(self.item_encoder, torch.nn.Embedding) and len(self.item_fields) == 1
                and type(self.score_func) in (InnerProductScorer, CosineScorer, EuclideanScorer)
                and (self.sampler is None or type(self.sampler) in (UniformSampler, PopularSamplerModel))
                and self.config['train'].get('sampling_method', 'none') == 'none'
                and not isinstance(getattr(self, 'neg_count', None), (list, tuple)))

    def forward(self, batch: Dict, full_score: bool = False, return_query: bool = False, return_item: bool = False,
                return_neg_item: bool = False, return_neg_id: bool = False):
        if not self._fused_ok():
            return self._forward_plugins(batch, full_score, return_query, return_item, return_neg_item, return_neg_id)
        output = {}
        pos_items = self._get_item_feat(batch)
        cosine = self.score_func.cosine      # False / True / rsa_score_mode of the Euclidean scorer
        query = None
        if self.sampler is not None:
            if self.neg_count is None:
                raise ValueError('`negative_count` value is required when `sampler` is not none.')
            qfeat = self._get_query_feat(batch)
            if isinstance(self.query_encoder, torch.nn.Embedding) and isinstance(qfeat, torch.Tensor) \
                    and qfeat.dim() == 1 and pos_items.dim() == 1 and not return_query:
                qsrc, qidx = self.query_encoder.weight, qfeat          # user-row gather fused into the kernel
            else:
                query = self.query_encoder(qfeat)
                qsrc, qidx = query.reshape(-1, query.shape[-1]), None
            lead = tuple(pos_items.shape)
            score, neg_ids = retriever_scores(
                self.item_encoder.weight, qsrc, self.neg_count, query_index=qidx, pos_ids=pos_items.reshape(-1),
                sampler=self.sampler, cosine=cosine, mask_pad_pos=pos_items.dim() > 1,
                sparse_grad=self.config['train'].get('sparse_grad', False))
            n = self.neg_count
            output['score'] = {
                'pos_score': score['pos_score'].view(lead), 'log_pos_prob': score['log_pos_prob'].view(lead).detach(),
                'neg_score': score['neg_score'].view(*lead, n), 'log_neg_prob': score['log_neg_prob'].view(*lead, n).detach()}
            neg_ids = neg_ids.view(*lead, n)
            if return_neg_item:
                output['neg_item'] = self.item_encoder(neg_ids)
            if return_neg_id:
                output['neg_id'] = neg_ids
        else:
            query = self.query_encoder(self._get_query_feat(batch))
            pos_score = self.score_func(query, self.item_encoder(pos_items))
            if batch[self.fiid].dim() > 1:
                pos_score = pos_score.masked_fill(batch[self.fiid] == 0, -float('inf'))
            output['score'] = {'pos_score': pos_score}
            if full_score:
                output['score']['all_score'] = self.score_func(query, self._get_item_vector())
        if return_query:
            output['query'] = query
        if return_item:
            output['item'] = self.item_encoder(pos_items)
        return output

    def _forward_plugins(self, batch, full_score, return_query, return_item, return_neg_item, return_neg_id):
        """baseretriever.py:142-192 step by step: sampler -> item_encoder -> score_func."""
        output = {}
        pos_items = self._get_item_feat(batch)
        pos_item_vec = None
        if self.sampler is not None:
            (log_pos_prob, neg_item_idx, log_neg_prob), query = self.sampling(
                batch=batch, num_neg=self.neg_count, excluding_hist=self.config['train'].get('excluding_hist', False),
                method=self.config['train'].get('sampling_method', 'none'), return_query=True)
            if (isinstance(self.item_encoder, torch.nn.Embedding) and len(self.item_fields) == 1
                    and type(self.score_func) in (InnerProductScorer, CosineScorer, EuclideanScorer)
                    and isinstance(pos_items, torch.Tensor) and isinstance(neg_item_idx, torch.Tensor)
                    and neg_item_idx.shape[:-1] == pos_items.shape and not return_neg_item and query.is_cuda):
                # Any sampler plugin / sampling method over an nn.Embedding catalog with a stock scorer: its ids go to
                # the gather+score kernel as GIVEN ids -- no [B, n, d] tensor of negative rows, and the backward is the
                # sorted row scatter instead of torch's atomic embedding backward (MaskedUniformSampler with
                # excluding_hist, B = 16384, n = 64: step + backward 3.1 -> 0.6 ms)
                n = neg_item_idx.shape[-1]
                score, _ = retriever_scores(
                    self.item_encoder.weight, query.reshape(-1, query.shape[-1]), n, pos_ids=pos_items.reshape(-1),
                    neg_ids=neg_item_idx.reshape(-1, n), cosine=self.score_func.cosine, mask_pad_pos=pos_items.dim() > 1,
                    sparse_grad=self.config['train'].get('sparse_grad', False))
                pos_score = score['pos_score'].view(pos_items.shape)
                neg_score = score['neg_score'].view(*pos_items.shape, n)
            else:
                pos_item_vec = self.item_encoder(pos_items)
                pos_score = self.score_func(query, pos_item_vec)
                if batch[self.fiid].dim() > 1:
                    pos_score = pos_score.masked_fill(batch[self.fiid] == 0, -float('inf'))
                neg_item_vec = self.item_encoder(self._get_item_feat(neg_item_idx))
                neg_score = self.score_func(query, neg_item_vec)
                if return_neg_item:
                    output['neg_item'] = neg_item_vec
            output['score'] = {'pos_score': pos_score, 'log_pos_prob': log_pos_prob, 'neg_score': neg_score,
                               'log_neg_prob': log_neg_prob}
            if return_neg_id:
                output['neg_id'] = neg_item_idx
        else:
            pos_item_vec = self.item_encoder(pos_items)
            query = self.query_encoder(self._get_query_feat(batch))
            pos_score = self.score_func(query, pos_item_vec)
            if batch[self.fiid].dim() > 1:
                pos_score = pos_score.masked_fill(batch[self.fiid] == 0, -float('inf'))
            output['score'] = {'pos_score': pos_score}
            if full_score:
                output['score']['all_score'] = self.score_func(query, self._get_item_vector())
        if return_query:
            output['query'] = query
        if return_item:
            output['item'] = pos_item_vec if pos_item_vec is not None else self.item_encoder(pos_items)
        return output

    def _sample(self, batch, neg: int = 1, excluding_hist: bool = False, return_query: bool = True):
        query = self.query_encoder(self._get_query_feat(batch))
        pos_items = batch.get(self.fiid, None)
        user_hist = batch.get('user_hist', batch.get(self.fiid, None)) if excluding_hist else None
        if not isinstance(self.sampler, Sampler):
            raise TypeError('`sampler` only support Sampler type.')
        kwargs = {'num_neg': neg, 'pos_items': pos_items}
        params = inspect.signature(self.sampler.forward).parameters      # baseretriever.py:225-228
        if 'excluding_hist' in params:
            kwargs['excluding_hist'] = excluding_hist
        if 'user_hist' in params:
            kwargs['user_hist'] = user_hist
        kwargs['query'] = query
        pos_prob, neg_id, neg_prob = self.sampler(**kwargs)
        return (pos_prob, neg_id, neg_prob, query) if return_query else (pos_prob, neg_id, neg_prob)

    def _pool_scores(self, query, pool):
        """Scores of every query against its own pool of item ids [.., n0] with the gather+score kernel (ids
        given, no [B, n0, d] tensor) -- no autograd, as in the reference's sampling() (which detaches)."""
        if not (isinstance(self.item_encoder, torch.nn.Embedding)
                and type(self.score_func) in (InnerProductScorer, CosineScorer, EuclideanScorer)):
            raise NotImplementedError("this sampling_method needs an nn.Embedding item tower and a stock scorer")
        n0 = pool.shape[-1]
        with torch.no_grad():
            q2 = query.reshape(-1, query.shape[-1]).contiguous()
            s = ops.fused_forward(self.item_encoder.weight, q2, n0, neg_ids=pool.reshape(-1, n0).contiguous(),
                                  cosine=self.score_func.cosine)['neg_score']
        return s.view(*pool.shape)

    def sampling(self, batch, num_neg, method='none', excluding_hist=False, t=1, return_query=False, query=None):
        """baseretriever.py:248-369, all six methods.  'none' is the sampler plugin; 'dns' / 'sir' (:330-355) draw a
        pool of num_neg[0] ids, score it with the gather+score kernel and keep the num_neg[1] best (`rsa_row_topk`) /
        resample by softmax weight; 'toprand' / 'top&rand' (:280-298) start from the full-catalog top-k (MFMA
        kernel); 'brute' (:300-328) samples from the softmax over the whole catalog.  The random draws of the
        non-default methods are the reference's own torch ops on the device (torch.randint / torch.multinomial),
        so they consume the generator exactly as the reference does."""
        if method not in ('none', 'dns', 'sir', 'toprand', 'top&rand', 'brute'):
            raise NotImplementedError('sampling method only support one of none/brute/sir/dns/toprand/top&rand')
        if isinstance(num_neg, int):
            num_neg = [num_neg, num_neg]
        elif not isinstance(num_neg, (list, tuple)):
            raise TypeError('num_neg only support int and List/Tuple type.')
        assert len(num_neg) == 2 and num_neg[0] >= num_neg[1], 'negative_count must be [pool, kept] with pool >= kept'
        fiid_val = batch.get(self.fiid)
        user_hist = batch.get('user_hist', fiid_val)
        if user_hist is not None and user_hist.dim() == 1:
            user_hist = user_hist.view(-1, 1)          # the positives play the history's role (:259-261)
        if method == 'none':
            assert self.sampler is not None, 'excepted sampler of retriever to be Sampler, but get None.'
            log_pos_prob, neg_id, log_neg_prob, query = self._sample(batch, num_neg[1], excluding_hist, True)
        elif method in ('dns', 'sir'):
            assert self.sampler is not None, 'excepted sampler of retriever to be Sampler, but get None.'
            _, pool, _, query = self._sample(batch, num_neg[0], excluding_hist, True)
            scores = self._pool_scores(query, pool)
            if method == 'dns':
                _, cols = ops.row_topk(scores, num_neg[1])
                neg_id = torch.gather(pool, -1, cols)
                log_neg_prob = torch.zeros_like(neg_id)
                log_pos_prob = torch.zeros_like(fiid_val)
            else:
                with torch.no_grad():
                    log_pos_prob = self.score_func(query, self.item_encoder(self._get_item_feat(batch)))      # :349-351
                    probs = torch.softmax(scores + torch.finfo(torch.float32).eps, dim=-1)
                    resampled = torch.multinomial(probs.reshape(-1, probs.shape[-1]), num_neg[1],
                                                  replacement=True).view(*probs.shape[:-1], num_neg[1])
                neg_id = torch.gather(pool, -1, resampled)
                log_neg_prob = torch.gather(scores, -1, resampled)
        elif method == 'toprand':
            _, topk_items, query = self.topk(batch, k=num_neg[0], user_h=user_hist, return_query=True)
            rand_idx = torch.randint(0, num_neg[0], (topk_items.size(0), num_neg[1]), device=topk_items.device)
            neg_id = torch.gather(topk_items, -1, rand_idx)
            log_neg_prob = torch.zeros_like(neg_id)
            log_pos_prob = torch.zeros_like(fiid_val)
        elif method == 'top&rand':
            k0 = num_neg[1] // 2
            _, neg_id, query = self.topk(batch, k=k0, user_h=user_hist, return_query=True)
            n_q = int(np.prod(query.shape[:-1]))
            rand_ids = torch.randint(1, self.item_vector.size(0) + 1, size=(n_q, num_neg[1] - k0), device=query.device)
            neg_id = torch.cat((neg_id, rand_ids), dim=-1)
            log_neg_prob = torch.zeros_like(neg_id)
            log_pos_prob = torch.zeros_like(fiid_val)
        else:                                                        # 'brute': softmax over the whole catalog
            query = self.query_encoder(self._get_query_feat(batch)) if query is None else query
            pos2 = fiid_val.view(-1, 1) if fiid_val.dim() == 1 else fiid_val
            with torch.no_grad():
                all_score = self.score_func(query, self.item_vector) / t
                all_prob = torch.nn.functional.pad(torch.softmax(all_score, dim=-1), pad=(1, 0))
                log_pos_prob = torch.log(torch.gather(all_prob, dim=-1, index=pos2))
                sampling_prob = all_prob
                if excluding_hist:
                    sampling_prob = all_prob.scatter(-1, user_hist, 0.0)        # utils.mask_with_hist(dist, hist, 0)
                neg_id = torch.multinomial(sampling_prob, num_neg[1] * pos2.size(-1), replacement=True)
                log_neg_prob = torch.log(torch.gather(sampling_prob, dim=-1, index=neg_id))
        log_pos_prob = log_pos_prob.view_as(fiid_val)
        result = (log_pos_prob.detach(), neg_id, log_neg_prob.detach())
        return (result, query) if return_query else (result, None)

    def topk(self, batch, k, user_h=None, return_query=False):
        """baseretriever.py:374-397: full-catalog scores -> top (k + |hist|) -> drop history -> top k."""
        query = self.query_encoder(self._get_query_feat(batch))
        if getattr(self, '_shard', None) is not None:
            return self._topk_sharded(query, k, user_h, return_query)
        more = user_h.size(1) if user_h is not None else 0
        if type(self.score_func) in (InnerProductScorer, CosineScorer, EuclideanScorer) \
                and isinstance(self.item_encoder, torch.nn.Embedding):
            # the MFMA kernel: scores + exact top-(k + more) without the [B, N] matrix (cosine / Euclidean: the tile
            # epilogue applies the norms, scorer.py:19-34); embed_dim > 128 or k + more > 1024 (a user whose history
            # is longer than that, e.g. ml-1m) are composed from the same kernel on materialised scores (ops.fullscore)
            n_items = self.item_encoder.weight.shape[0]
            kc = min(k + more, n_items - 1)
            table = self.item_vector if hasattr(self, 'item_vector') else self._get_item_vector()
            q = query.detach().contiguous()
            in_kernel = min(512, n_items - 1)     # a selection the threshold-filter pass of the kernel still makes (k <= 656)
            if user_h is not None and kc > 1024 and 2 * k <= in_kernel and table.shape[1] <= 128:
                # A history longer than 1024 - k (ml-1m: ~1.8 k) pushes k + |hist| past the in-kernel select and onto
                # materialised scores + torch.topk (25 ms instead of 5.5 at B = 2048, N = 1e6).  The k best items
                # outside the history are among the 512 best overall unless more than 512 - k history items outrank
                # them: select 512 in the kernel, drop the history, and only the rows left with fewer than k survivors
                # (heavy users of a model that ranks their own history first) go through the wide path.
                _, _, score, topk_items = ops.fullscore(table.detach(), q, k=in_kernel, items_without_pad=True,
                                                        score_mode=self.score_func.cosine)
                score, topk_items = ops.topk_mask_history(score, topk_items, user_h, k)
                short = torch.isinf(score[:, k - 1]).nonzero().view(-1)
                if short.numel():
                    _, _, s2, i2 = ops.fullscore(table.detach(), q[short].contiguous(), k=kc, items_without_pad=True,
                                                 score_mode=self.score_func.cosine)
                    s2, i2 = ops.topk_mask_history(s2, i2, user_h[short].contiguous(), k)
                    score[short], topk_items[short] = s2, i2
                return (score, topk_items, query) if return_query else (score, topk_items)
            _, _, score, topk_items = ops.fullscore(table.detach(), q, k=kc,
                                                    items_without_pad=True, score_mode=self.score_func.cosine)
        else:
            score, topk_items = torch.topk(self.score_func(query, self.item_vector), k + more)
            topk_items = topk_items + 1
        if user_h is not None:
            score, topk_items = ops.topk_mask_history(score, topk_items, user_h, min(k, score.shape[1]))
        return (score, topk_items, query) if return_query else (score, topk_items)

    # ------------------------------------------------------------------ steps
    def training_step(self, batch):
        # single-launch path: stock BPRLoss on the fused kernels with a whole number of 64-negative tiles
        if (type(self.loss_fn) is BPRLoss and self.sampler is not None and self._fused_ok()
                and type(self.score_func) is InnerProductScorer and self.neg_count and self.neg_count % 64 == 0
                and batch[self.fiid].dim() == 1):
            qfeat = self._get_query_feat(batch)
            if isinstance(self.query_encoder, torch.nn.Embedding) and isinstance(qfeat, torch.Tensor) and qfeat.dim() == 1:
                qsrc, qidx = self.query_encoder.weight, qfeat
            else:
                qsrc, qidx = self.query_encoder(qfeat), None
            loss, _ = fused_bpr_loss(self.item_encoder.weight, qsrc, self.neg_count, query_index=qidx,
                                     pos_ids=batch[self.fiid], sampler=self.sampler,
                                     sparse_grad=self.config['train'].get('sparse_grad', False))
            return loss
        # the same for stock SampledSoftmaxLoss (configs[2]): the logsumexp over the positive and the negatives of a
        # query is carried through its tiles by one wave, the query gradient is accumulated in the forward
        if (type(self.loss_fn) is SampledSoftmaxLoss and self.sampler is not None and self._fused_ok()
                and type(self.score_func) is InnerProductScorer and self.neg_count and self.neg_count % 64 == 0
                and batch[self.fiid].dim() == 1 and self.item_encoder.weight.shape[1] in (32, 64, 128, 256)
                and self.config['train'].get('fused_ssm', True)):
            qfeat = self._get_query_feat(batch)
            if isinstance(self.query_encoder, torch.nn.Embedding) and isinstance(qfeat, torch.Tensor) and qfeat.dim() == 1:
                qsrc, qidx = self.query_encoder.weight, qfeat
            else:
                qsrc, qidx = self.query_encoder(qfeat), None
            loss, _ = fused_ssm_loss(self.item_encoder.weight, qsrc, self.neg_count, query_index=qidx,
                                     pos_ids=batch[self.fiid], sampler=self.sampler,
                                     sparse_grad=self.config['train'].get('sparse_grad', False))
            return loss
        # full softmax without the [B, N] score matrix: stock SoftmaxLoss over an nn.Embedding catalog
        if (type(self.loss_fn) is SoftmaxLoss and self.sampler is None and type(self.score_func) is InnerProductScorer
                and isinstance(self.item_encoder, torch.nn.Embedding) and len(self.item_fields) == 1
                and self.item_encoder.weight.shape[1] <= 128 and batch[self.fiid].dim() == 1
                and self.config['train'].get('fused_full_softmax', True)):
            output = self.forward(batch, False, return_query=True)
            lse = full_lse(output['query'], self.item_encoder.weight)
            return (lse - output['score']['pos_score']).mean()       # loss_func.py:39-47, first branch
        output = self.forward(batch, isinstance(self.loss_fn, FullScoreLoss))
        score = output['score']
        score['label'] = batch[self.frating]
        return self.loss_fn(**score)

    def validation_step(self, batch):
        cutoff = self.config['eval']['cutoff']
        cutoff = cutoff[0] if isinstance(cutoff, list) else cutoff
        return self._test_step(batch, self.config['eval']['val_metrics'], [cutoff])

    def test_step(self, batch):
        cutoff = self.config['eval']['cutoff']
        return self._test_step(batch, self.config['eval']['test_metrics'], cutoff if isinstance(cutoff, list) else [cutoff])

    def _test_step(self, batch, metric, cutoffs):
        rank_m = rs_eval.get_rank_metrics(metric)
        topk = self.config['eval']['topk']
        bs = batch[self.frating].size(0)
        assert len(rank_m) > 0
        score, topk_items = self.topk(batch, topk, batch['user_hist'])
        n_valid = batch.get('_n_valid')
        if n_valid is not None and n_valid < bs:
            # this rank's part of a global evaluation batch the world size did not divide: the trailing rows are padding
            # (dataset.rank_part) -- they took part in the (collective) top-k above and are dropped from the metrics
            if n_valid == 0:
                return {f'{name}@{cutoff}': 0.0 for cutoff in cutoffs for name, _ in rank_m}, 0
            bs = n_valid
            topk_items = topk_items[:bs]
            batch = dict(batch, **{self.fiid: batch[self.fiid][:bs], self.frating: batch[self.frating][:bs]})
        if batch[self.fiid].dim() > 1:
            target, _ = batch[self.fiid].sort()
            idx_ = torch.searchsorted(target, topk_items)
            idx_[idx_ == target.size(1)] = target.size(1) - 1
            label = torch.gather(target, 1, idx_) == topk_items
            pos_rating = batch[self.frating]
        else:
            label = batch[self.fiid].view(-1, 1) == topk_items
            pos_rating = batch[self.frating].view(-1, 1)
        return {f'{name}@{cutoff}': func(label, pos_rating, cutoff) for cutoff in cutoffs for name, func in rank_m}, bs

    # ------------------------------------------------------------------ minimal fit / evaluate loop
    def _device(self):
        if not torch.cuda.is_available():
            raise RuntimeError('recstudio_amd trains on a ROCm GPU only (no CPU fallback)')
        gpu = self.config['train'].get('gpu', 1)
        idx = gpu[0] if isinstance(gpu, (list, tuple)) and gpu else 0
        return torch.device('cuda', int(idx) if isinstance(idx, int) and idx < torch.cuda.device_count() else 0)

    def _get_optimizer(self):
        tr = self.config['train']
        name, lr, wd = tr['learner'].lower(), tr['learning_rate'], tr['weight_decay']
        params = [p for p in self.parameters() if p.requires_grad]
        if tr.get('sparse_grad', False):
            return torch.optim.SparseAdam(params, lr=lr) if name == 'adam' else torch.optim.SGD(params, lr=lr)
        table = {'adam': torch.optim.Adam, 'sgd': torch.optim.SGD, 'adagrad': torch.optim.Adagrad,
                 'rmsprop': torch.optim.RMSprop}
        return table.get(name, torch.optim.Adam)(params, lr=lr, weight_decay=wd)

    def _get_scheduler(self, optimizer):
        """recommender.py:476-494: ``train.scheduler`` = 'exponential' (gamma 0.98) | 'onplateau' | None, stepped once
        per epoch (the plateau scheduler on the monitored validation metric, else on the training loss)."""
        name = self.config['train'].get('scheduler')
        if name is None or optimizer is None:
            return None
        if name.lower() == 'exponential':
            return torch.optim.lr_scheduler.ExponentialLR(optimizer, gamma=0.98)
        if name.lower() == 'onplateau':
            return torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode=self.config['train'].get('early_stop_mode', 'max'))
        return None

    def _step_scheduler(self, scheduler, log):
        if scheduler is None:
            return
        if isinstance(scheduler, torch.optim.lr_scheduler.ReduceLROnPlateau):
            scheduler.step(log.get(getattr(self, 'val_metric', None), log['train_loss']))
        else:
            scheduler.step()
        log['lr'] = float(scheduler.optimizer.param_groups[0]['lr'])

    def _to_device(self, batch, device):
        def move(v):
            if isinstance(v, torch.Tensor):
                return v.to(device, non_blocking=True)
            if isinstance(v, tuple):
                return tuple(move(x) for x in v)
            return v
        return {k: move(v) for k, v in batch.items()}

    # ------------------------------------------------------------------ multi-GPU fit / evaluate (row-sharded item table)
    def _dist_context(self, kwargs):
        """The process group this ``fit`` runs in when it is ONE RANK of a multi-process job: ``fit(..., dist=...)`` (a
        torch.distributed look-alike; forces the sharded path at any world size), an initialised torch.distributed
        with more than one rank, or a launcher environment (WORLD_SIZE > 1: the group is created here, RCCL, one rank
        per GPU).  None: single-process training.  Replaces recommender.py:608-612, :716-742 (``data_parallel`` /
        ``_accelerate``: every parameter re-broadcast to a gpu list each step; the ddp branch raises)."""
        dist = kwargs.get('dist')
        if dist is not None:
            return dist
        import torch.distributed as td
        if not td.is_available():
            return None
        if not td.is_initialized() and int(os.environ.get('WORLD_SIZE', '1')) > 1:
            local = int(os.environ.get('LOCAL_RANK', '0'))
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            torch.cuda.set_device(local)
            td.init_process_group('nccl', device_id=torch.device('cuda', local))
        if td.is_initialized() and td.get_world_size() > 1:
            return td
        gpu = self.config['train'].get('gpu', 1)
        n_gpu = len(gpu) if isinstance(gpu, (list, tuple)) else (gpu if isinstance(gpu, int) else 1)
        if n_gpu > 1:
            # the reference would now replicate the model over the gpu list inside this process (accelerator 'dp') or raise
            # ('ddp'); here multi-GPU training is one process per GPU
            self.logger.warning(f"train.gpu asks for {n_gpu} GPUs but this is a single process: training on one GPU.  Start the "
                                f"same script with `python -m torch.distributed.run --nproc-per-node {n_gpu} --master-addr "
                                f"127.0.0.1 script.py` and fit() shards the item table over the ranks.")
        return None

    def _setup_shard(self, train_data, dist, backend=None, device=None):
        """Row-shard the item table over the ranks of ``dist``: rank r keeps rows [lo, hi) -- with ``train.shard_layout:
        'interleaved'`` rows r, r + G, ... -- of ``item_encoder.weight`` (and,
        through the optimizer, of its Adam state) as an ``nn.Embedding`` of its own block; sampler and query tower are
        replicated.  Every rank initialises the full model from the same seed first (recommender.py:34-35), so the
        shards are the rows of the very table a single process would train."""
        from . import shard
        rank, world = dist.get_rank(), dist.get_world_size()
        if not (isinstance(self.item_encoder, torch.nn.Embedding) and len(self.item_fields) == 1
                and type(self.score_func) is InnerProductScorer and isinstance(self.loss_fn, PairwiseLoss)
                and self.sampler is not None and self.config['train'].get('sampling_method', 'none') == 'none'
                and isinstance(self.neg_count, int) and self.neg_count > 0):
            raise NotImplementedError('multi-GPU fit covers the two-tower configuration: nn.Embedding item tower over the item '
                                      'id, InnerProductScorer, a PairwiseLoss, a Sampler with sampling_method "none"')
        if device is None:
            if backend is not None:
                raise ValueError('a non-default shard backend needs an explicit device')
            device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', torch.cuda.current_device())))
        device = torch.device(device)
        n_items, d = self.item_encoder.weight.shape
        plan = shard.RowShardPlan(n_items, world, layout=self.config['train'].get('shard_layout', 'block'))
        if self.item_encoder.weight.is_meta:
            rows = _device_init_block(plan, rank, n_items, d, self.config['train']['seed'] or 2022,
                                      self.config['train']['init_method'], device)
            if rank == 0 and self.item_encoder.padding_idx is not None:
                rows[self.item_encoder.padding_idx] = 0
        else:
            rows = plan.take(self.item_encoder.weight.detach(), rank).clone()
        block = torch.nn.Embedding(plan.n_local(rank), d, padding_idx=0 if rank == 0 else None, _weight=rows)
        # An ITEM-TOWER query encoder (SASRec: seq/sasrec.py:14, :42, :107) embeds its input with the very table the
        # model scores against.  Every place the tower holds that module is re-pointed at the sharded table below
        # (shard.ShardedRows: ids out, rows back, gradients to the owners), so the embedding stays tied and no rank
        # keeps -- or dense-all-reduces the gradient of -- a replica of the catalog.
        full = self.item_encoder
        tied = []
        if isinstance(self.query_encoder, torch.nn.Module):
            tied = [(mod, name) for mod in self.query_encoder.modules() for name, child in mod._modules.items() if child is full]
            for mod, name in tied:
                mod._modules[name] = None
            if any(p is full.weight for p in self.query_encoder.parameters()):
                for mod, name in tied:
                    mod._modules[name] = full
                raise NotImplementedError('multi-GPU fit: the query tower uses item_encoder.weight outside the item_encoder '
                                          'module itself; only towers that call their item_encoder submodule can be sharded')
        self.item_encoder = block                      # the full table is dropped: this rank keeps its rows only
        lo, hi = (None, None) if plan.interleaved else plan.bounds(rank)
        if hasattr(self, 'item_vector'):
            del self.item_vector
        self.to(device)
        table = shard.ShardedItemTable(self.item_encoder.weight.data, plan, rank, dist, backend=backend,
                                       sample_seed=self.config['train']['seed'] or 2022,
                                       chunks=int(self.config['train'].get('shard_slices', 1)),
                                       owner_loss=bool(self.config['train'].get('shard_owner_loss', True)),
                                       deterministic=bool(self.config['train'].get('shard_deterministic', True)),
                                       rows_share=self.config['train'].get('shard_rows_share'))
        self._shard = {'table': table, 'dist': dist, 'rank': rank, 'world': world, 'lo': lo, 'hi': hi, 'device': device,
                       'n_items': n_items, 'plan': plan, 'tower_rows': None}
        if tied:
            rows = shard.ShardedRows(table)
            for mod, name in tied:
                mod._modules[name] = rows
            self._shard['tower_rows'] = rows
        return self._shard

    def _topk_sharded(self, query, k, user_h, return_query):
        """``topk`` over the sharded catalog: per-shard MFMA pass + exchange of the (value, id) partials
        (ShardedItemTable.full_lse_topk), then the history exclusion of baseretriever.py:386-392.  The candidate count is
        k + the width of the WHOLE history table (the same on every rank: the collectives need equal shapes) -- at least
        as many candidates as the single-process path keeps, so the k survivors are the same."""
        sh = self._shard
        table, be = sh['table'], sh['table'].backend
        more = sh.get('hist_width', user_h.size(1) if user_h is not None else 0)
        kc = min(k + more, sh['n_items'] - 1)
        mask = be.mask_history if hasattr(be, 'mask_history') else ops.topk_mask_history
        q = query.detach().contiguous()
        narrow = min(sh.get('topk_narrow', 512), sh['n_items'] - 1)
        if user_h is not None and kc > ops.FULLSCORE_MAX_K and 2 * k <= narrow:
            # a history longer than 1024 - k (ml-1m: ~1.8 k): the single-process strategy (``topk`` above) with
            # rank-uniform shapes -- 512 candidates over the whole catalog, history dropped; only if ANY row of ANY rank is
            # left with fewer than k survivors (one flag, all-reduced: every rank takes the same branch) does the batch
            # go through the wide pass
            _, score, topk_items = table.full_lse_topk(q, narrow, want_lse=False)
            score, topk_items = mask(score, topk_items, user_h, k)
            short = torch.isinf(score[:, k - 1]).any().to(torch.int32).reshape(1)
            table._all_reduce_max(short)
            if not int(short):
                return (score, topk_items, query) if return_query else (score, topk_items)
        _, score, topk_items = table.full_lse_topk(q, kc, want_lse=False)
        if user_h is not None:
            score, topk_items = mask(score, topk_items, user_h, min(k, score.shape[1]))
        else:
            score, topk_items = score[:, :k], topk_items[:, :k]
        return (score, topk_items, query) if return_query else (score, topk_items)

    def _prepared_batches(self, loader, device):
        for batch in loader:
            batch = self._to_device(batch, device)
            batch.pop('_n_valid', None)
            yield batch

    def _fit_sharded(self, train_data, val_data, dist, backend=None, device=None):
        """One process per GPU, item table row-sharded, query tower replicated (``shard.ShardedRetriever``): rank r trains on
        its contiguous 1/G of every global batch (``train.batch_size`` is per rank), the negatives come from one job-wide
        stream, the loss is the global mean -- a G-rank run reproduces the single-process run on the global batch up to
        fp32 summation order.  The optimizer is the reference's (``train.learner``) on this rank's parameters: its own
        row block (with its block of the optimizer state) and its replica of the tower, whose gradients are summed over
        the ranks first; ``train.fused_optimizer: 'sgd'`` applies plain SGD inside the exchange instead."""
        from . import shard
        # train.shard_init: 'host' = every rank initialises the full table as a single process would and keeps its rows
        # (bit-equal to the unsharded model's initial weights); 'device' = the rows are drawn on the owner's device, the
        # full table never exists anywhere (_device_init_block); 'auto' (default): 'device' once the table exceeds 2 GiB
        mode = self.config['train'].get('shard_init', 'auto')
        self._item_table_on_meta = mode == 'device' or (mode == 'auto' and not self.item_encoder and
                                                          train_data.num_items * self.embed_dim * 4 > (2 << 30))
        self._init_model(train_data)
        self._init_parameter()
        sh = self._setup_shard(train_data, dist, backend, device)
        table, rank, world, device = sh['table'], sh['rank'], sh['world'], sh['device']
        if hasattr(self.sampler, 'to'):
            self.sampler.to(device)
        if getattr(type(self.sampler), 'update', Sampler.update) is not Sampler.update:
            raise NotImplementedError('multi-GPU fit: samplers with a per-epoch update() are not covered')
        if val_data is not None:
            val_data.use_field = train_data.use_field
        tr = self.config['train']
        if hasattr(train_data, 'user_hist') and train_data.user_hist is not None:
            sh['hist_width'] = int(max(train_data.user_hist.shape[1], getattr(val_data, 'user_hist', train_data.user_hist).shape[1]))
        fused = tr.get('fused_optimizer')
        if fused not in (None, 'sgd'):
            raise NotImplementedError("multi-GPU fit: train.fused_optimizer must be None or 'sgd'")
        lr = tr['learning_rate']
        embed_tower = isinstance(self.query_encoder, torch.nn.Embedding)
        trainer = shard.ShardedRetriever(table, self.query_encoder, self.sampler, self.loss_fn, self.neg_count,
                                         item_sgd_lr=lr if fused else None,
                                         query_sgd_lr=lr if (fused and embed_tower) else None)
        optimizer = None
        if not fused or not embed_tower:
            params = [p for p in self.parameters() if p.requires_grad and not (fused and p is self.item_encoder.weight)]
            optimizer = self._make_optimizer(params) if params else None
        if not fused:
            self.item_encoder.weight.grad = trainer.item_grad_local       # the exchange accumulates into the block's .grad
        sched_opt = optimizer
        if fused and embed_tower and tr.get('scheduler'):      # no torch optimizer at all: a stand-in carries the rate
            sched_opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=lr)
            sched_opt.step()                                   # (a no-op; the schedulers warn when they step first)
        scheduler = self._get_scheduler(sched_opt)
        sh['trainer'] = trainer
        trainer.table.defer_overflow = True          # report a capacity overflow after the step instead of raising inside it
        if sh['tower_rows'] is not None:
            sh['tower_rows'].bind(trainer)
        val_metrics = self.config['eval']['val_metrics']
        cutoff = self.config['eval']['cutoff']
        cutoff0 = cutoff[0] if isinstance(cutoff, list) else cutoff
        self.val_metric = f"{(val_metrics[0] if isinstance(val_metrics, list) else val_metrics)}@{cutoff0}"
        on_gpu = device.type == 'cuda'
        best, best_state, bad = None, None, 0
        self.train_losses, self.history = [], []
        for epoch in range(tr['epochs']):
            t0 = time.time()
            self.train()
            losses = []
            # train.shard_lookahead: the weight-independent half of the NEXT batch's step (negatives, routing, key exchange,
            # the owner's sorts) is issued on a second stream before the current batch is stepped (ShardedRetriever.prepare_step)
            ahead = bool(tr.get('shard_lookahead', False)) and trainer.can_prepare()
            self._shard['lookahead'] = ahead
            if on_gpu and tr.get('device_loader', True) and hasattr(train_data, 'device_train_loader'):
                loader = train_data.device_train_loader(tr['batch_size'], shuffle=True, drop_last=False, device=device, ddp=True,
                                                        rank=rank, world=world)
                # the rank parts of one global batch: every rank's history window has the same shape, so a tower's row
                # look-ups can travel in fixed-capacity segments (no host round trip per step).  NOT with look-ahead: a
                # fixed-capacity look-up hands its dropped count to the NEXT routing launch, which with look-ahead belongs to a
                # batch prepared one or two steps earlier on the other stream -- the step whose look-up overflowed would run
                # ungated.  The variable-split look-up cannot drop anything.
                trainer.table.uniform_lookups = not ahead
            else:
                loader = train_data.train_loader(batch_size=tr['batch_size'], shuffle=True, drop_last=False, ddp=True,
                                                 rank=rank, world=world)
            ticket = None
            with _above_second_stream(ahead and on_gpu, device, self._shard):
                for batch, batch_next in _with_next(self._prepared_batches(loader, device)):
                    if ahead and ticket is None:
                        ticket = trainer.prepare_step(self._get_query_feat(batch), batch[self.fiid])
                    ticket_next = trainer.prepare_step(self._get_query_feat(batch_next), batch_next[self.fiid]) \
                        if ahead and batch_next is not None else None
                    if optimizer is not None:
                        optimizer.zero_grad(set_to_none=False)
                    loss = trainer.training_step(self._get_query_feat(batch), batch[self.fiid], batch[self.frating], ticket=ticket)
                    ticket = ticket_next
                    if optimizer is not None:
                        if tr['grad_clip_norm'] is not None:
                            self._clip_grad_norm_sharded(params, tr['grad_clip_norm'], dist)
                        optimizer.step()
                    losses.append(loss.detach().reshape(1))
                    # an id distribution that outgrew the calibrated segment capacity: the poll (every rank sees the job-wide
                    # sticky count at the same step) reports it HERE, after the whole step -- the steps in which elements were
                    # dropped updated nothing (their scale was 0 on the device), the step the lagging poll landed in is a healthy
                    # one and has run to its end, the capacity is recalibrated on the next step; carry on
                    msg = trainer.table.take_overflow()
                    if msg:
                        self.logger.warning(msg)
            step_losses = torch.cat(losses)                     # this rank's shares of the global mean losses
            dist.all_reduce(step_losses)
            self.train_losses.append(step_losses.cpu())
            log = {'epoch': epoch, 'train_loss': float(step_losses.mean()), 'train_time': time.time() - t0}
            if val_data is not None and (epoch + 1) % self.config['eval']['val_n_epoch'] == 0:
                log.update(self._eval_epoch(val_data, self.validation_step, device))
                cur = log[self.val_metric]
                better = best is None or (cur > best if tr['early_stop_mode'] == 'max' else cur < best)
                if better:
                    best, best_state, bad = cur, copy.deepcopy(self.state_dict()), 0
                else:
                    bad += 1
            self._step_scheduler(scheduler, log)      # (the all-reduced loss / metric: the same decision on every rank)
            if fused and scheduler is not None:
                trainer.set_sgd_lr(log['lr'])
            self.logged_metrics = log
            self.history.append(dict(log))
            if rank == 0:
                self.logger.info(' '.join(f'{k}={v:.4f}' if isinstance(v, float) else f'{k}={v}' for k, v in log.items()))
            if val_data is not None and bad >= tr['early_stop_patience']:
                break
        if best_state is not None:
            self.load_state_dict(best_state)
            table.item_local = self.item_encoder.weight.data
        return best

    def _clip_grad_norm_sharded(self, params, max_norm, dist):
        """``torch.nn.utils.clip_grad_norm_`` (recommender.py training loop) over a model whose item table is split over the
        ranks: the squared norm of this rank's row block is summed over the ranks, the replicated tower's gradients
        (already summed, identical everywhere) count once; every rank scales by the same coefficient."""
        block = self.item_encoder.weight
        sq_block = torch.zeros((), device=block.device)
        sq_rest = torch.zeros((), device=block.device)
        for p in params:
            if p.grad is None:
                continue
            if p is block:
                sq_block = sq_block + p.grad.float().pow(2).sum()
            else:
                sq_rest = sq_rest + p.grad.float().pow(2).sum()
        dist.all_reduce(sq_block)
        total = (sq_block + sq_rest).sqrt()
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        for p in params:
            if p.grad is not None:
                p.grad.mul_(coef)
        return total

    def _make_optimizer(self, params):
        tr = self.config['train']
        name, lr, wd = tr['learner'].lower(), tr['learning_rate'], tr['weight_decay']
        table = {'adam': torch.optim.Adam, 'sgd': torch.optim.SGD, 'adagrad': torch.optim.Adagrad,
                 'rmsprop': torch.optim.RMSprop}
        return table.get(name, torch.optim.Adam)(params, lr=lr, weight_decay=wd)

    def fit(self, train_data, val_data=None, run_mode='light', config: Dict = None, **kwargs):
        if config is not None:
            self.config.update(config)
        dist = self._dist_context(kwargs)
        if dist is not None:
            return self._fit_sharded(train_data, val_data, dist, kwargs.get('shard_backend'), kwargs.get('device'))
        self._init_model(train_data)
        self._init_parameter()
        device = self._device()
        self.to(device)
        if val_data is not None:
            val_data.use_field = train_data.use_field
        optimizer = self._get_optimizer()
        tr = self.config['train']
        fused_step = self._fused_optimizer_step(tr)
        sched_opt = optimizer
        if fused_step is not None:
            # train.scheduler with a fused optimizer: the reference's scheduler objects drive a stand-in optimizer whose
            # learning rate is handed to the kernels after every epoch
            sched_opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=tr['learning_rate'])
            sched_opt.step()                                   # (a no-op; the schedulers warn when they step first)
        scheduler = self._get_scheduler(sched_opt)
        val_metrics = self.config['eval']['val_metrics']
        cutoff = self.config['eval']['cutoff']
        cutoff0 = cutoff[0] if isinstance(cutoff, list) else cutoff
        self.val_metric = f"{(val_metrics[0] if isinstance(val_metrics, list) else val_metrics)}@{cutoff0}"
        best, best_state, bad = None, None, 0
        self.history = []                                  # one dict per epoch: what the reference logs (recommender.py:249-270)
        self._fused_step = fused_step
        # the device-resident loader is built once: its columns are uploaded once, every epoch re-draws the order
        dev_loader = None
        if tr.get('device_loader', True) and hasattr(train_data, 'device_train_loader'):
            dev_loader = train_data.device_train_loader(tr['batch_size'], shuffle=True, drop_last=False, device=device)
        for epoch in range(tr['epochs']):
            t0 = time.time()
            self.train()
            self._update_item_vector()
            if self.sampler is not None:
                self.sampler.update(item_embs=self.item_vector)                 # recommender.py:564-570
            losses = []
            if dev_loader is not None:
                loader = dev_loader
            else:
                loader = train_data.train_loader(batch_size=tr['batch_size'], shuffle=True, drop_last=False)
            if getattr(fused_step, 'stepper', None) is not None:
                # one batch of look-ahead: the next step's negatives are drawn and sorted on a side stream while this
                # step's forward and apply passes run (fused.PrefetchedBPRSGD; same weights bit for bit)
                stepper, ticket = fused_step.stepper, None
                for batch in loader:
                    batch = self._to_device(batch, device)
                    nxt = stepper.prepare(batch[self.fuid], batch[self.fiid])
                    if ticket is not None:
                        losses.append(stepper.step(ticket)[0])
                    ticket = nxt
                if ticket is not None:
                    losses.append(stepper.step(ticket)[0])
                loader = ()
            for batch in loader:
                batch = self._to_device(batch, device)
                if fused_step is not None and batch[self.fiid].dim() == 1:
                    losses.append(fused_step(batch))        # forward + loss + optimizer update in the kernels
                    continue
                optimizer.zero_grad()
                loss = self.training_step(batch)
                loss.backward()
                if tr['grad_clip_norm'] is not None:
                    torch.nn.utils.clip_grad_norm_(self.parameters(), tr['grad_clip_norm'])
                optimizer.step()
                losses.append(loss.detach())
            log = {'epoch': epoch, 'train_loss': float(torch.stack(losses).mean()), 'train_time': time.time() - t0}
            if val_data is not None and (epoch + 1) % self.config['eval']['val_n_epoch'] == 0:
                t1 = time.time()
                log.update(self._eval_epoch(val_data, self.validation_step, device))
                log['valid_time'] = time.time() - t1              # the reference logs both (recommender.py:262-266)
                cur = log[self.val_metric]
                better = best is None or (cur > best if tr['early_stop_mode'] == 'max' else cur < best)
                if better:
                    best, best_state, bad = cur, copy.deepcopy(self.state_dict()), 0
                else:
                    bad += 1
            self._step_scheduler(scheduler, log)
            if fused_step is not None and scheduler is not None:
                fused_step.set_lr(log['lr'])
            self.logged_metrics = log
            self.history.append(dict(log))
            self.logger.info(' '.join(f'{k}={v:.4f}' if isinstance(v, float) else f'{k}={v}' for k, v in log.items()))
            if val_data is not None and bad >= tr['early_stop_patience']:
                break
        if best_state is not None:
            self.load_state_dict(best_state)
        return best

    def _fused_optimizer_step(self, tr):
        """``train.fused_optimizer: 'sgd' | 'adam'`` (default None = the reference's torch optimizer on autograd
        gradients): for the stock BPR two-tower configuration (nn.Embedding towers, InnerProductScorer, BPRLoss,
        Uniform / Popular sampler, negative_count % 64 == 0, embed_dim in {64, 128, 256}) run the whole step --
        sampling, scoring, loss, and the SGD / lazy-Adam update of the touched rows -- in the kernels, with no gradient
        tensors (fused.bpr_sgd_step / fused.FusedBPRAdam).  'adam' follows torch.optim.SparseAdam (rows outside the
        batch keep their state), which differs from the dense Adam the reference runs by default."""
        kind = tr.get('fused_optimizer')
        if not kind:
            return None
        ok = (type(self.loss_fn) is BPRLoss and self.sampler is not None and self._fused_ok()
              and type(self.score_func) is InnerProductScorer and isinstance(self.neg_count, int) and self.neg_count % 64 == 0
              and isinstance(self.query_encoder, torch.nn.Embedding) and self.item_encoder.weight.shape[1] in (64, 128, 256)
              and not tr.get('weight_decay') and tr.get('grad_clip_norm') is None)
        if not ok:
            raise NotImplementedError("train.fused_optimizer needs the stock BPR two-tower configuration "
                                      "(see BaseRetriever._fused_optimizer_step)")
        from .fused import FusedBPRAdam, bpr_sgd_step
        iw, uw, lr = self.item_encoder.weight, self.query_encoder.weight, tr['learning_rate']
        if kind == 'sgd':
            rate = {'lr': lr}

            def sgd_step(b):
                return bpr_sgd_step(iw, uw, self.neg_count, rate['lr'], user_ids=b[self.fuid], pos_ids=b[self.fiid],
                                    sampler=self.sampler)[0]
            sgd_step.stepper = None
            if self.neg_count == 64 and tr.get('fused_prefetch', True) and self.fiid is not None:
                from .fused import PrefetchedBPRSGD
                sgd_step.stepper = PrefetchedBPRSGD(iw, uw, self.neg_count, lr, self.sampler)

            def set_lr(new):
                rate['lr'] = float(new)
                if sgd_step.stepper is not None:
                    sgd_step.stepper.set_lr(new)
            sgd_step.set_lr = set_lr
            return sgd_step
        if kind == 'adam':
            opt = FusedBPRAdam(iw, uw, lr=lr)

            def adam_step(b):
                return opt.step(self.neg_count, user_ids=b[self.fuid], pos_ids=b[self.fiid], sampler=self.sampler)[0]
            adam_step.stepper = None
            # (one batch of look-ahead does not pay here -- in-process A/B at the headline shape: 2.61 ms plain, 2.68 ms one
            # batch ahead: the draw leaves the forward for the stand-alone sampler and the apply pass is the step -- so it is
            # opt-in for the lazy-Adam step: train.fused_prefetch: 'adam')
            if tr.get('fused_prefetch', True) == 'adam':
                class _Ahead:      # the prepare / step pair the fit loop drives one batch ahead
                    @staticmethod
                    def prepare(uid, pos):
                        return opt.prepare(self.neg_count, user_ids=uid, pos_ids=pos, sampler=self.sampler)
                    step = staticmethod(opt.step_prepared)
                adam_step.stepper = _Ahead
            adam_step.set_lr = lambda new: setattr(opt, 'lr', float(new))
            return adam_step
        raise ValueError(f"train.fused_optimizer must be 'sgd' or 'adam', got {kind!r}")

    @torch.no_grad()
    def _eval_epoch(self, data, step, device):
        self.eval()
        sh = getattr(self, '_shard', None)
        if sh is None:
            self._update_item_vector()
            loader = data.eval_loader(batch_size=self.config['eval']['batch_size'])
        else:       # this rank's contiguous part of every global evaluation batch (dataset.py:1141-1142)
            loader = data.eval_loader(batch_size=self.config['eval']['batch_size'], ddp=True, rank=sh['rank'], world=sh['world'])
        total, rows, sizes = 0, [], []
        for batch in self._eval_batches(data, loader, device, cache=sh is None):
            metrics, bs = step(batch)
            # per-batch values stay on the device until the epoch ends: float(v) here was one host synchronisation per
            # metric per batch (most of an ml-100k validation epoch)
            rows.append(metrics)
            sizes.append(bs)
            total += bs
        acc = {}
        if rows:
            keys = list(rows[0])
            vals = torch.stack([torch.stack([torch.as_tensor(m[k], dtype=torch.float32, device=device) for k in keys])
                                for m in rows]).double().cpu()
            for r, bs in enumerate(sizes):                                         # weighted mean, recommender.py:308-324
                for c, k in enumerate(keys):
                    acc[k] = acc.get(k, 0.0) + float(vals[r, c]) * bs
        if sh is not None:                                                     # sums over the ranks' parts
            keys = sorted(acc)
            t = torch.tensor([acc[k] for k in keys] + [float(total)], dtype=torch.float64, device=device)
            if t.device.type == 'cuda':
                t = t.float()
            sh['dist'].all_reduce(t)
            acc, total = {k: float(v) for k, v in zip(keys, t[:-1])}, float(t[-1])
        return {k: v / max(total, 1) for k, v in acc.items()}

    EVAL_CACHE_BYTES = 1 << 30

    def _eval_batches(self, data, loader, device, cache=True):
        """The evaluation batches of ``data`` on the device.  An evaluation split does not change between epochs, so the
        host loader's work (per-user ragged slices, ``pad_sequence``, the history rows, the uploads) is done once: the
        device-resident batches of a split are kept (up to EVAL_CACHE_BYTES over all splits) and replayed every epoch."""
        store = self.__dict__.setdefault('_eval_cache', {})
        key = (id(data), self.config['eval']['batch_size'], str(device), len(data))
        hit = store.get(key)
        if hit is not None and hit[0] is data:
            yield from hit[1]
            return
        kept, size = [], 0
        budget = self.EVAL_CACHE_BYTES - sum(v[2] for v in store.values())
        for batch in loader:
            batch = self._to_device(batch, device)
            if cache and kept is not None:
                size += sum(v.numel() * v.element_size() for v in batch.values() if isinstance(v, torch.Tensor))
                if size > budget:
                    kept = None
                else:
                    kept.append(batch)
            yield batch
        if cache and kept is not None:
            store[key] = (data, kept, size)

    def evaluate(self, test_data, verbose=True, **kwargs):
        test_data.use_field = self.fields
        sh = getattr(self, '_shard', None)
        if sh is not None and getattr(test_data, 'user_hist', None) is not None:
            sh['hist_width'] = max(sh.get('hist_width', 0), int(test_data.user_hist.shape[1]))
        out = self._eval_epoch(test_data, self.test_step, next(self.parameters()).device)
        if verbose:
            self.logger.info(str(out))
        return out


TwoTowerRecommender = BaseRetriever      # README.md:29-31 names
ItemTowerRecommender = BaseRetriever


class BPR(BaseRetriever):
    """recstudio/model/mf/bpr.py:7-25, recstudio/model/mf/config/bpr.yaml (negative_count: 1)."""

    def __init__(self, config=None, **kwargs):
        super().__init__(config, **kwargs)
        if 'negative_count' not in (config or {}).get('train', {}):
            self.config['train']['negative_count'] = 1

    def _get_dataset_class():
        return TripletDataset

    def _get_item_encoder(self, train_data):
        return torch.nn.Embedding(train_data.num_items, self.embed_dim, padding_idx=0)

    def _get_query_encoder(self, train_data):
        return torch.nn.Embedding(train_data.num_users, self.embed_dim, padding_idx=0)

    def _get_score_func(self):
        return InnerProductScorer()

    def _get_loss_func(self):
        return BPRLoss()

    def _get_sampler(self, train_data):
        return UniformSampler(train_data.num_items)


class _EmbedFn(torch.autograd.Function):
    """item_encoder(ids) with the HIP gather forward and the HIP row scatter-add backward."""

    @staticmethod
    def forward(ctx, weight, ids):
        ctx.save_for_backward(ids)
        ctx.n_rows = weight.shape[0]
        return ops.embedding_gather(weight, ids)

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        return _embedding_grad(g, ids, ctx.n_rows), None


class _above_second_stream:
    """``with`` block whose work runs on a HIGH-priority stream (entered after, left before, the stream that was current):
    the look-ahead's second stream then sits below the steps it runs under and fills their gaps instead of competing with
    them (sharded training step one batch ahead, in process: 1.166 ms with both at the default priority, 1.136 ms so)."""

    def __init__(self, on, device, cache):
        self.on, self.device, self.cache = bool(on), device, cache

    def __enter__(self):
        if self.on:
            self.prev = torch.cuda.current_stream(self.device)
            hi = self.cache.get('hi_stream')
            if hi is None:
                hi = self.cache['hi_stream'] = torch.cuda.Stream(device=self.device, priority=-1)
            hi.wait_stream(self.prev)
            torch.cuda.set_stream(hi)
            self.hi = hi
        return self

    def __exit__(self, *exc):
        if self.on:
            self.prev.wait_stream(self.hi)
            torch.cuda.set_stream(self.prev)
        return False


def _with_next(it):
    """(x0, x1), (x1, x2), ..., (x_last, None)"""
    it = iter(it)
    cur = next(it, None)
    while cur is not None:
        nxt = next(it, None)
        yield cur, nxt
        cur = nxt


def _embedding_grad(g, ids, n_rows):
    """Dense ``weight.grad`` of an item-row gather (embedding_dense_backward with padding_idx = 0): rows sorted by item id
    and summed run by run without atomics (``rsa_rows_update_sorted``; bit-reproducible) for the stock dims, the
    float-atomic scatter otherwise."""
    d = g.shape[-1]
    rows = g.reshape(-1, d).contiguous()
    flat_ids = ids.reshape(-1, 1).contiguous()
    if d in (64, 128, 256) and rows.shape[0] > 0:
        ones = torch.ones(rows.shape[0], 1, dtype=torch.float32, device=g.device)
        return ops.scatter_rows_sorted(torch.zeros(n_rows, d, dtype=torch.float32, device=g.device), rows, flat_ids, ones, pad_row=0)
    return ops.scatter_add_rows(rows, flat_ids.view(-1), n_rows)


class _SegEmbedFn(torch.autograd.Function):
    """item_encoder(in_item_id) straight from the CSR view of the interaction column (dataset.py:1418-1439 +
    seq/sasrec.py:42): ``rsa_seg_gather`` reads ``[start, end)`` of the flat item column and writes the right-padded
    ``[B, L, d]`` rows -- no ``[B, L]`` id tensor is re-read; the backward is the sorted, atomics-free row scatter."""

    @staticmethod
    def forward(ctx, weight, flat, start, end, max_len, ids):
        ctx.save_for_backward(ids)
        ctx.n_rows = weight.shape[0]
        return ops.seg_gather(weight, flat, start, end, int(max_len), want_rows=True, want_ids=False)[1]

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        return _embedding_grad(g, ids, ctx.n_rows), None, None, None, None, None


class SASRecQueryEncoder(torch.nn.Module):
    """recstudio/model/seq/sasrec.py:8-67: item-embedding gather of the history (HIP) + learned positions
    + causal nn.TransformerEncoder (stock PyTorch-ROCm, out of scope) + last-position pooling."""

    def __init__(self, fiid, embed_dim, max_seq_len, n_head, hidden_size, dropout, activation, layer_norm_eps, n_layer,
                 item_encoder):
        super().__init__()
        self.fiid = fiid
        self.item_encoder = item_encoder
        self.position_emb = torch.nn.Embedding(max_seq_len, embed_dim)
        layer = torch.nn.TransformerEncoderLayer(d_model=embed_dim, nhead=n_head, dim_feedforward=hidden_size,
                                                 dropout=dropout, activation=activation, layer_norm_eps=layer_norm_eps,
                                                 batch_first=True, norm_first=False)
        self.transformer_layer = torch.nn.TransformerEncoder(layer, num_layers=n_layer)
        self.dropout = torch.nn.Dropout(p=dropout)

    def forward(self, batch):
        user_hist = batch['in_' + self.fiid]
        B, L = user_hist.shape
        positions = torch.arange(L, dtype=torch.long, device=user_hist.device).unsqueeze(0).expand(B, L)
        seg = batch.get('_seg')
        if not isinstance(self.item_encoder, torch.nn.Embedding):
            # the row-sharded catalog (shard.ShardedRows put here by BaseRetriever._setup_shard): ids out, rows back
            rows = self.item_encoder(user_hist)
        elif seg is not None and user_hist.is_cuda:
            # the device loader handed over the CSR view (flat item column, start, end): the fused segment gather emits
            # [B, L, d] directly; ``in_item_id`` only serves the padding mask and the backward
            flat, start, end = seg
            rows = _SegEmbedFn.apply(self.item_encoder.weight, flat, start, end, L, user_hist)
        else:
            rows = _EmbedFn.apply(self.item_encoder.weight, user_hist)
        seq = rows + self.position_emb(positions)
        causal = torch.triu(torch.ones(L, L, dtype=torch.bool, device=user_hist.device), 1)
        out = self.transformer_layer(self.dropout(seq), mask=causal, src_key_padding_mask=user_hist == 0)
        last = (batch['seqlen'] - 1).clamp(min=0).view(-1, 1, 1).expand(-1, 1, out.shape[-1])
        return out.gather(1, last).squeeze(1)


class SASRec(BaseRetriever):
    """recstudio/model/seq/sasrec.py:70-123 with the retriever tail on the fused path."""

    def __init__(self, config=None, **kwargs):
        # recstudio/model/seq/config/sasrec.yaml on top of basemodel.yaml; the caller's config wins
        user = config or {}
        super().__init__(config, **kwargs)
        m = self.config['model']
        for k, v in (('hidden_size', 128), ('layer_num', 2), ('head_num', 2), ('dropout_rate', 0.5),
                     ('activation', 'gelu'), ('layer_norm_eps', 1e-12)):
            m.setdefault(k, v)
        for k, v in (('negative_count', 1), ('init_method', 'normal')):
            if k not in user.get('train', {}):
                self.config['train'][k] = v

    def _get_dataset_class():
        return SeqDataset

    def _get_query_encoder(self, train_data):
        m = self.config['model']
        return SASRecQueryEncoder(self.fiid, self.embed_dim, train_data.config['max_seq_len'], m['head_num'],
                                  m['hidden_size'], m['dropout_rate'], m['activation'], m['layer_norm_eps'],
                                  m['layer_num'], self.item_encoder)

    def _get_query_feat(self, data):
        return data if isinstance(data, dict) else super()._get_query_feat(data)

    def _get_loss_func(self):
        return BinaryCrossEntropyLoss()         # sasrec.py:117-119

    def _get_sampler(self, train_data):
        return UniformSampler(train_data.num_items)
