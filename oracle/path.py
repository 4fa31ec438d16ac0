"""Oracle (test infrastructure): CPU restatement of the RecStudio retriever hot path.

Plain PyTorch-CPU / numpy restatement of what the reference computes between
``BaseRetriever.forward`` and the loss, each function citing the reference
file:line it follows (paths relative to /root/reference).  It is checked
against fixtures recorded from the real reference (oracle/make_golden.py ->
tests/golden/*.npz).  Never imported by the product package.
"""
import math

import numpy as np
import torch

from . import philox


# --------------------------------------------------------------------------- samplers
class UniformSampler:
    """recstudio/ann/sampler.py:81-114 (+ base class :48-58).

    ``num_items`` is the table height N (padding row 0 included); ids are drawn
    from [1, N-1].  ``forward`` on CPU consumes torch's CPU generator exactly
    as the reference does; ``forward_device_stream`` restates what the same
    call produces on a ROCm device for generator state (seed, offset).
    """

    def __init__(self, num_items):
        self.num_items = num_items - 1                       # sampler.py:51

    @staticmethod
    def _shape(query):
        if isinstance(query, int):                           # sampler.py:91-94
            return (query,)
        return tuple(query.shape[:-1])                       # sampler.py:95-99

    def forward(self, query, num_neg, pos_items=None):
        shape = self._shape(query)
        nq = int(np.prod(shape))
        neg = torch.randint(1, self.num_items + 1, size=(nq, num_neg))   # sampler.py:102-104
        neg = neg.reshape(*shape, -1)                                   # :105
        neg_prob = torch.zeros_like(neg)                                # :113-114 (int64 zeros)
        if pos_items is not None:
            return torch.zeros_like(pos_items), neg, neg_prob
        return neg, neg_prob

    def forward_device_stream(self, query, num_neg, seed, offset, grid_threads, pos_items=None):
        shape = self._shape(query)
        nq = int(np.prod(shape))
        ids = philox.device_randint(seed, offset, nq * num_neg, 1, self.num_items + 1, grid_threads)
        neg = torch.from_numpy(ids).reshape(*shape, num_neg)
        neg_prob = torch.zeros_like(neg)
        if pos_items is not None:
            return torch.zeros_like(pos_items), neg, neg_prob
        return neg, neg_prob


def popular_tables(pop_count, mode=0):
    """recstudio/ann/sampler.py:224-241: (pop_prob, table) fp32 [N] built with torch CPU ops."""
    pc = torch.as_tensor(pop_count).to(torch.float)
    if mode == 0:
        pc = torch.log(pc + 1)                               # :229-230
    elif mode == 1:
        pc = torch.log(pc + 1) + 1e-6                        # :231-232
    elif mode == 2:
        pc = pc ** 0.75                                      # :233-234
    pc[0] = 1                                                # :237  (padding id is sampleable)
    pop_prob = pc / pc.sum()                                 # :239
    table = torch.cumsum(pop_prob, dim=0)                    # :240
    pop_prob = pop_prob.clone()
    pop_prob[-1] = 1.0                                       # :241  (after the table is built)
    return pop_prob, table


def searchsorted_left(table, u):
    """First index i with table[i] >= u (== torch.searchsorted(table, u), sampler.py:247)."""
    t = np.asarray(table, dtype=np.float32)
    return np.searchsorted(t, np.asarray(u, dtype=np.float32), side='left').astype(np.int64)


class PopularSamplerModel:
    """recstudio/ann/sampler.py:224-258 (inverse-CDF popularity sampling)."""

    def __init__(self, pop_count, mode=0):
        self.num_items = len(pop_count) - 1
        self.pop_prob, self.table = popular_tables(pop_count, mode)

    def compute_item_p(self, ids):
        return torch.log(self.pop_prob[ids])                 # :257-258

    def ids_from_uniform(self, u, clamp=True):
        ids = torch.from_numpy(searchsorted_left(self.table.numpy(), np.asarray(u)))
        if clamp:
            # the reference would index out of range when u > table[-1]; the build clamps
            ids = ids.clamp_(max=self.table.numel() - 1)
        return ids

    def forward(self, query, num_neg, pos_items=None):
        shape = tuple(query.shape[:-1])
        nq = int(np.prod(shape))
        seeds = torch.rand(nq, num_neg)                      # :246
        neg = torch.searchsorted(self.table, seeds)          # :247
        neg = neg.reshape(*shape, -1)                        # :249
        neg_prob = self.compute_item_p(neg)
        if pos_items is not None:
            return self.compute_item_p(pos_items), neg, neg_prob
        return neg, neg_prob

    def forward_device_stream(self, query, num_neg, seed, offset, grid_threads, pos_items=None):
        shape = tuple(query.shape[:-1])
        nq = int(np.prod(shape))
        u = philox.device_rand(seed, offset, nq * num_neg, grid_threads)
        neg = self.ids_from_uniform(u).reshape(*shape, num_neg)
        neg_prob = self.compute_item_p(neg)
        if pos_items is not None:
            return self.compute_item_p(pos_items), neg, neg_prob
        return neg, neg_prob


def masked_uniform_from_u(num_items, num_neg, user_hist, u, num_query_per_user=None):
    """recstudio/ann/sampler.py:117-147 with the uniforms given: rejection-free uniform sampling over the
    items NOT in each user's (0-padded) history.  ``num_items`` excludes the padding id."""
    n_q = 1 if num_query_per_user is None else num_query_per_user
    num_user, hist_len = user_hist.shape
    u = torch.as_tensor(u, dtype=torch.float32).view(num_user, n_q * num_neg)
    nz = torch.count_nonzero(user_hist, dim=-1)
    neg = torch.floor(u * (num_items - nz).view(-1, 1)).long() + 1                 # :133
    sorted_hist, _ = user_hist.sort(dim=-1)
    offset = torch.arange(hist_len).repeat(num_user, 1) - (hist_len - nz).view(-1, 1)
    offset[offset < 0] = 0
    sorted_hist = sorted_hist - offset                                             # :134-138
    masked_offset = torch.searchsorted(sorted_hist, neg, right=True)               # :139
    neg = neg + (masked_offset - (hist_len - nz).view(-1, 1))                      # :140-141
    if num_query_per_user is not None:
        neg = neg.reshape(num_user, num_query_per_user, num_neg)
    return neg


# --------------------------------------------------------------------------- scorers
def inner_product_score(query, items):
    """recstudio/model/scorer.py:5-17 -- dispatch on shapes exactly as the reference."""
    if query.size(0) == items.size(0):
        if query.dim() < items.dim():                        # ([B,D],[B,n,D]) / ([B,L,D],[B,L,n,D])
            out = torch.matmul(items, query.unsqueeze(-1)).squeeze(-1)
        else:                                                # ([B,D],[B,D]) / ([B,L,D],[B,L,D])
            out = (query * items).sum(-1)
    else:                                                    # ([B,D],[N,D])
        out = query @ items.T
    return out


def cosine_score(query, items):
    """recstudio/model/scorer.py:19-25 -- no epsilon; zero rows give NaN/inf like the reference."""
    out = inner_product_score(query, items)
    out = out / torch.norm(items, dim=-1)
    keep = (query.dim() != items.dim()) or (query.size(0) != items.size(0))
    out = out / torch.norm(query, dim=-1, keepdim=keep)
    return out


def euclidean_score(query, items):
    """recstudio/model/scorer.py:28-34: -(-2 <q, x> + |x|^2 + |q|^2)."""
    out = -2 * inner_product_score(query, items)
    out = out + torch.sum(torch.square(items), dim=-1)
    keep = (query.dim() != items.dim()) or (query.size(0) != items.size(0))
    out = out + torch.sum(torch.square(query), dim=-1, keepdim=keep)
    return -out


# --------------------------------------------------------------------------- losses
def norm_score(query, items, p=2):
    """recstudio/model/scorer.py:56-66 (NormScorer): minus the p-norm of (query - items)."""
    if query.dim() < items.dim() or query.size(0) != items.size(0):
        query = query.unsqueeze(-2)
    return -torch.norm(query - items, p=p, dim=-1)


def gmf_score(query, items, weight, bias=None, activation=torch.relu):
    """recstudio/model/scorer.py:69-86 (GMFScorer): act(Linear(query * item)); 2-D queries as in the reference."""
    if query.dim() < items.dim():
        query = query.unsqueeze(1)
    elif query.size(0) != items.size(0):
        query, items = query.unsqueeze(1), items.unsqueeze(0)
    return activation(torch.nn.functional.linear(query * items, weight, bias)).squeeze(-1)


def bpr_loss(pos_score, neg_score):
    """recstudio/model/loss_func.py:55-59 (dns=False)."""
    diff = pos_score.unsqueeze(-1) - neg_score
    ls = torch.nn.functional.logsigmoid(diff)
    w = torch.softmax(torch.ones_like(neg_score), -1)
    return -(ls * w).sum(-1).mean()


def sampled_softmax_loss(pos_score, log_pos_prob, neg_score, log_neg_prob):
    """recstudio/model/loss_func.py:80-90."""
    new_pos = pos_score - log_pos_prob
    new_neg = neg_score - log_neg_prob
    if new_pos.dim() < new_neg.dim():
        new_pos = new_pos.unsqueeze(-1)
    cat = torch.cat([new_pos, new_neg], dim=-1)
    out = torch.logsumexp(cat, dim=-1, keepdim=True) - new_pos
    notpad = torch.logical_not(torch.isinf(new_pos)).float().sum(-1)
    out = torch.nan_to_num(out, posinf=0).sum(-1) / notpad
    return out.mean()


def bce_loss(pos_score, neg_score):
    """recstudio/model/loss_func.py:105-127 (dns=False; weight = 1/n, :129-130)."""
    weight = torch.ones_like(neg_score) / neg_score.size(-1)
    pad = torch.isinf(pos_score)
    pos_loss = torch.nn.functional.logsigmoid(pos_score).masked_fill(pad, 0.0).sum() / (~pad).sum()
    neg_loss = (torch.nn.functional.softplus(neg_score) * weight).sum(-1)
    if pos_score.dim() == neg_score.dim() - 1:
        neg_loss = neg_loss.masked_fill(pad, 0.0).sum() / (~pad).sum()
    else:
        neg_loss = neg_loss.mean()
    return -pos_loss + neg_loss


def _neg_weights(neg_score, log_neg_prob):
    """softmax over the negatives of (score - log q): the importance weights of the "Weighted" losses
    (recstudio/model/loss_func.py:96, :136-137).  Part of the autograd graph, as in the reference."""
    return torch.softmax(neg_score - log_neg_prob, dim=-1)


def weighted_bpr_loss(pos_score, neg_score, log_neg_prob):
    """recstudio/model/loss_func.py:93-97."""
    ls = torch.nn.functional.logsigmoid(pos_score.unsqueeze(-1) - neg_score)
    return -(ls * _neg_weights(neg_score, log_neg_prob)).sum(-1).mean()


def weighted_bce_loss(pos_score, neg_score, log_neg_prob):
    """recstudio/model/loss_func.py:135-137 on top of :105-127 (dns=False)."""
    pad = torch.isinf(pos_score)
    n_valid = (~pad).sum()
    pos_term = torch.nn.functional.logsigmoid(pos_score).masked_fill(pad, 0.0).sum() / n_valid
    neg_rows = (torch.nn.functional.softplus(neg_score) * _neg_weights(neg_score, log_neg_prob)).sum(-1)
    if pos_score.dim() == neg_score.dim() - 1:
        neg_term = neg_rows.masked_fill(pad, 0.0).sum() / n_valid
    else:
        neg_term = neg_rows.mean()
    return neg_term - pos_term


def hinge_loss(pos_score, neg_score, margin=2.0):
    """recstudio/model/loss_func.py:140-154 with num_items=None (the num_items branch takes the mean of a bool
    tensor, which torch rejects)."""
    hardest = neg_score.max(dim=-1).values
    return torch.clamp(hardest - pos_score + margin, min=0).mean()


def info_nce_loss(pos_score, neg_score):
    """recstudio/model/loss_func.py:157-160: sampled softmax with the proposal log-probabilities zeroed."""
    return sampled_softmax_loss(pos_score, torch.zeros_like(pos_score), neg_score, torch.zeros_like(neg_score))


def nce_loss(pos_score, log_pos_prob, neg_score, log_neg_prob):
    """recstudio/model/loss_func.py:163-168 (x - softplus(x) written as logsigmoid(x) -- the same function)."""
    zp, zn = pos_score - log_pos_prob, neg_score - log_neg_prob
    per_row = torch.nn.functional.logsigmoid(zp) + (zn - torch.nn.functional.softplus(zn)).sum(1)
    return -per_row.mean()


def ccl_loss(pos_score, neg_score, margin=0.8, neg_weight=0.3):
    """recstudio/model/loss_func.py:171-186."""
    p, q = torch.sigmoid(pos_score), torch.sigmoid(neg_score)
    pushed = torch.relu(q - margin).mean(dim=-1)
    n_valid = torch.logical_not(torch.isinf(p)).float().sum()
    return torch.nan_to_num((1 - p) + neg_weight * pushed, posinf=0.0).sum() / n_valid


def softmax_loss(pos_score, all_score):
    """recstudio/model/loss_func.py:39-47."""
    if all_score.dim() > pos_score.dim():
        return (torch.logsumexp(all_score, dim=-1) - pos_score).mean()
    out = torch.logsumexp(all_score, dim=-1, keepdim=True) - pos_score
    notpad = torch.logical_not(torch.isinf(pos_score)).float().sum(-1)
    out = torch.nan_to_num(out, posinf=0).sum(-1) / notpad
    return out.mean()


# --------------------------------------------------------------------------- retriever forward
def retriever_forward(item_w, query, pos_ids, neg_ids, cosine=False, full=False):
    """recstudio/model/basemodel/baseretriever.py:142-192 for an Embedding item tower.

    item_w [N,d] (row 0 = padding), query [B,d] or [B,L,d], pos_ids [B] / [B,L],
    neg_ids [B,n] / [B,L,n].  Returns (pos_score, neg_score[, all_score])."""
    score = cosine_score if cosine else inner_product_score
    pos_vec = item_w[pos_ids]                                # :153-154
    pos_score = score(query, pos_vec)                        # :163
    if pos_ids.dim() > 1:
        pos_score = pos_score.masked_fill(pos_ids == 0, -float('inf'))   # :164-165
    out = [pos_score]
    if neg_ids is not None:
        neg_vec = item_w[neg_ids]                            # :167-168
        out.append(score(query, neg_vec))                    # :169
    if full:
        out.append(score(query, item_w[1:]))                 # :183-186 (item_vector = weight[1:], :122-123)
    return tuple(out)


def dense_grads(item_w, user_w, user_ids, pos_ids, neg_ids, loss='bpr', log_pos_prob=None,
                log_neg_prob=None, cosine=False):
    """Loss + dense table grads through autograd, as recommender.py:636-639 produces them
    (nn.Embedding(padding_idx=0): row 0 receives no gradient)."""
    iw = item_w.clone().requires_grad_(True)
    uw = user_w.clone().requires_grad_(True)
    q = torch.nn.functional.embedding(user_ids, uw, padding_idx=0)
    score = cosine_score if cosine else inner_product_score
    pos = score(q, torch.nn.functional.embedding(pos_ids, iw, padding_idx=0))
    if loss == 'softmax':
        alls = score(q, iw[1:])
        val = softmax_loss(pos, alls)
        neg = None
    else:
        neg = score(q, torch.nn.functional.embedding(neg_ids, iw, padding_idx=0))
        if loss == 'bpr':
            val = bpr_loss(pos, neg)
        else:
            val = sampled_softmax_loss(pos, log_pos_prob, neg, log_neg_prob)
    val.backward()
    return val.detach(), pos.detach(), (None if neg is None else neg.detach()), iw.grad, uw.grad


# --------------------------------------------------------------------------- top-k + metrics
def topk_with_history(query, item_w, k, user_hist=None, cosine=False):
    """recstudio/model/basemodel/baseretriever.py:374-397 (no ANN index)."""
    score_fn = cosine_score if cosine else inner_product_score
    more = user_hist.size(1) if user_hist is not None else 0
    score, items = torch.topk(score_fn(query, item_w[1:]), k + more)     # :384
    items = items + 1                                                    # :385
    if user_hist is not None:
        existing, _ = user_hist.sort()                                   # :387
        idx = torch.searchsorted(existing, items)
        idx[idx == existing.size(1)] = existing.size(1) - 1
        score[torch.gather(existing, 1, idx) == items] = -float('inf')   # :390
        score, sel = score.topk(k)
        items = torch.gather(items, 1, sel)
    return score, items


def test_step_hits(target_ids, topk_items):
    """recstudio/model/basemodel/baseretriever.py:422-430: bool hit matrix [B, topk]."""
    if target_ids.dim() > 1:
        target, _ = target_ids.sort()
        idx = torch.searchsorted(target, topk_items)
        idx[idx == target.size(1)] = target.size(1) - 1
        return torch.gather(target, 1, idx) == topk_items
    return target_ids.view(-1, 1) == topk_items


test_step_hits.__test__ = False      # not a pytest test


def rank_metrics(pred, target, k):
    """recstudio/eval/__init__.py:9-165 for the 2-D (ranking) branch.
    pred bool [B, topk]; target ratings [B, T] (>0 = relevant)."""
    pred_k = pred[:, :k].float()
    count = (target > 0).sum(-1)
    out = {}
    out['recall'] = (pred_k.sum(-1) / count).mean()                                  # :27-30
    out['precision'] = (pred_k.sum(-1) / k).mean()                                   # :53-56
    ap = (pred_k.cumsum(-1) / torch.arange(1, k + 1).float()) * pred_k               # :97-101
    out['map'] = (ap.sum(-1) / torch.minimum(count, k * torch.ones_like(count))).mean()

    def dcg(p, kk):
        kk = min(kk, p.size(1))
        return (p[:, :kk] / torch.log2(torch.arange(kk).float() + 2.0).view(1, -1)).sum(-1)
    pd = dcg(pred.float(), k)                                                        # :122-128
    ideal = dcg(torch.sort((target > 0).float(), descending=True)[0], k)
    irr = torch.all(target <= np.finfo(np.float64).eps, dim=-1)
    pd = torch.where(irr, torch.zeros_like(pd), pd / torch.where(irr, torch.ones_like(ideal), ideal))
    out['ndcg'] = pd.mean()
    first = torch.zeros(pred.size(0))                                                # :143-150
    hit = pred[:, :k]
    anyhit = hit.any(-1)
    firstpos = hit.float().argmax(-1) + 1
    first = torch.where(anyhit, 1.0 / firstpos.float(), torch.zeros_like(first))
    out['mrr'] = first.mean()
    out['hit'] = anyhit.float().mean()                                               # :165
    return out


# --------------------------------------------------------------------------- sequence gather
def seq_gather(item_w, flat_item_ids, start, end, max_len):
    """recstudio/data/dataset.py:1418-1439 + seq/sasrec.py:42: ragged [start,end) slices of the
    user-sorted item column -> right-padded ids [B,L] (pad 0) -> embedding rows [B,L,d]."""
    B = len(start)
    lens = (np.asarray(end) - np.asarray(start)).astype(np.int64)
    L = int(max_len)
    ids = torch.zeros(B, L, dtype=torch.int64)
    for b in range(B):
        n = int(lens[b])
        ids[b, :n] = flat_item_ids[int(start[b]):int(end[b])]
    return ids, item_w[ids], torch.from_numpy(lens)


# --------------------------------------------------------------------------- SASRec query tower
def sasrec_query(state, hist, seqlen, n_head, hidden_size, n_layer, activation='gelu', layer_norm_eps=1e-12):
    """recstudio/model/seq/sasrec.py:37-67 (unidirectional, 'last' pooling, dropout 0): item rows of the right-padded
    history + learned positions -> causal nn.TransformerEncoder with the padding mask -> the output at position seqlen - 1.
    ``state``: the reference module's state dict (item_encoder.weight, position_emb.weight, transformer_layer.*) as
    tensors.  Stock torch modules on the CPU, as the reference runs them."""
    item_w, pos_w = state['item_encoder.weight'], state['position_emb.weight']
    d = item_w.shape[1]
    layer = torch.nn.TransformerEncoderLayer(d_model=d, nhead=n_head, dim_feedforward=hidden_size, dropout=0.0,
                                             activation=activation, layer_norm_eps=layer_norm_eps, batch_first=True,
                                             norm_first=False)                                            # :19-28
    enc = torch.nn.TransformerEncoder(layer, num_layers=n_layer)                                         # :29-32
    enc.load_state_dict({k[len('transformer_layer.'):]: v for k, v in state.items() if k.startswith('transformer_layer.')})
    enc.train()                     # (the training-mode code path of nn.TransformerEncoder; dropout is 0)
    L = hist.shape[1]
    seq = item_w[hist] + pos_w[torch.arange(L)].unsqueeze(0)                                              # :38-42
    causal = torch.triu(torch.ones(L, L, dtype=torch.bool), 1)                                           # :46-47
    out = enc(seq, mask=causal, src_key_padding_mask=hist == 0)                                          # :50-53
    last = (seqlen - 1).view(-1, 1, 1).expand(-1, 1, d)
    return out.gather(1, last).squeeze(1)                                                                # :55-60 ('last' pooling)
