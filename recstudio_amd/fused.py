"""The fused hot path as autograd nodes: sample -> gather -> score (-> loss) with a row-wise
gradient scatter-add backward.

``retriever_scores`` is what ``BaseRetriever.forward`` dispatches to when the item tower is an
``nn.Embedding`` (recstudio/model/basemodel/baseretriever.py:153-171): it returns the same
``pos_score / log_pos_prob / neg_score / log_neg_prob`` tensors plus ``neg_id`` without ever
materialising the [B, n, d] negative rows.  Its backward is ``rsa_fused_backward`` and yields what
autograd produces in the reference (dense ``weight.grad`` with row 0 untouched), or row-sparse
COO gradients when ``sparse_grad=True`` (the only workable form at N = 1e7..1e8).
"""
import ctypes

import torch

from . import _native as nat
from . import ops, rng
from ._native import ptr
from .sampler import PopularSamplerModel, Sampler, UniformSampler


def _sampler_kind(sampler):
    # exact types only: a subclass may override forward, and must then go through the plugin path
    if type(sampler) is UniformSampler:
        return nat.SAMPLER_UNIFORM
    if type(sampler) is PopularSamplerModel:
        return nat.SAMPLER_POPULAR
    return None


_POP_KEYS = ('table', 'pop_prob', 'guide', 'guide_log2', 'table_prob', 'cdf_lut', 'cdf_lines', 'lines_log2')


def _pop_kw(cfg):
    """the popularity sampler's tables out of a cfg dict, as ops.fused_forward keyword arguments"""
    return {k: cfg[k] for k in _POP_KEYS if cfg.get(k) is not None}


class _ScoreFn(torch.autograd.Function):
    """(item_weight, query_src) -> (pos_score, neg_score); non-differentiable extras ride in ctx.extras."""

    @staticmethod
    def forward(ctx, item_weight, query_src, cfg):
        out = ops.fused_forward(item_weight, query_src, cfg['num_neg'], query_index=cfg.get('query_index'),
                                pos_ids=cfg.get('pos_ids'), neg_ids=cfg.get('neg_ids'), sampler=cfg['sampler'],
                                cosine=cfg.get('cosine', False), mask_pad_pos=cfg.get('mask_pad_pos', False),
                                n_queries=cfg.get('n_queries'), **_pop_kw(cfg))
        cfg['out'] = out
        ctx.cfg = cfg
        ctx.save_for_backward(item_weight, query_src, out['neg_ids'])
        pos = out.get('pos_score')
        if pos is None:
            pos = item_weight.new_zeros(0)
        return pos, out['neg_score']

    @staticmethod
    def backward(ctx, gpos, gneg):
        cfg = ctx.cfg
        item_weight, query_src, neg_ids = ctx.saved_tensors
        qi, pos_ids = cfg.get('query_index'), cfg.get('pos_ids')
        sparse = cfg.get('sparse_grad', False)
        need_item = ctx.needs_input_grad[0]
        need_q = ctx.needs_input_grad[1]
        if gneg is None:
            gneg = torch.zeros_like(cfg['out']['neg_score'])
        if pos_ids is not None and (gpos is None or gpos.numel() == 0):
            gpos = torch.zeros(pos_ids.numel(), dtype=torch.float32, device=item_weight.device)
        # dense user-table gradient straight from the kernel (no [M, d] intermediate, no second launch)
        qtab = torch.zeros_like(query_src) if (need_q and qi is not None and not sparse) else None
        # dense item gradient of the inner-product scorer: sorted atomics-free scatter (rsa_rows_update_sorted) next to
        # a backward launch that only produces the query gradient; other cases: the backward kernel's own scatter
        cos = cfg.get('cosine', False)
        sorted_dense = (need_item and not sparse and (cos is False or cos == nat.SCORE_IP)
                        and item_weight.shape[1] in (64, 128, 256) and neg_ids.shape[1] % 64 == 0)
        item_grad = rows = qgrad = None
        gneg_c = gneg.contiguous()
        gpos_c = None if pos_ids is None else gpos.contiguous()
        if sorted_dense:
            item_grad = ops.scatter_rows_sorted(torch.zeros_like(item_weight), query_src, neg_ids, gneg_c, query_index=qi,
                                                pos_ids=pos_ids, dpos=gpos_c, pad_row=0)
        if not sorted_dense or need_q:
            ig, rows, qgrad = ops.fused_backward(
                item_weight, query_src, neg_ids, gneg_c, query_index=qi, pos_ids=pos_ids, dpos=gpos_c,
                dense_item_grad=need_item and not sparse and not sorted_dense, row_item_grad=need_item and sparse,
                want_query_grad=need_q and qtab is None, query_table_grad=qtab, cosine=cos)
            if not sorted_dense:
                item_grad = ig
        g_item = None
        if need_item:
            if sparse:
                M, n = neg_ids.shape
                pos_col = pos_ids.view(M, 1) if pos_ids is not None else neg_ids.new_zeros(M, 1)
                idx = torch.cat([pos_col, neg_ids], 1).reshape(1, -1)
                g_item = torch.sparse_coo_tensor(idx, rows, item_weight.shape)
            else:
                g_item = item_grad
        g_q = None
        if need_q:
            if qi is not None:        # query_src is a user table: embedding_dense_backward
                if sparse:
                    g_q = torch.sparse_coo_tensor(qi.view(1, -1), qgrad, query_src.shape)
                else:
                    g_q = qtab
            else:
                g_q = qgrad
        return g_item, g_q, None


def retriever_scores(item_weight, query_src, num_neg, *, query_index=None, pos_ids=None, sampler=None,
                     neg_ids=None, cosine=False, mask_pad_pos=False, sparse_grad=False):
    """Fused BaseRetriever.forward body.  ``query_src`` is either the [M, d] query vectors or a user
    table together with ``query_index`` [M].  ``sampler`` is a UniformSampler / PopularSamplerModel
    (ids drawn in-kernel) or None with ``neg_ids`` given.  Returns the reference's score dict
    (baseretriever.py:170-171) plus ``neg_id``."""
    M = query_index.numel() if query_index is not None else query_src.shape[0]
    cfg = {'num_neg': int(num_neg), 'query_index': query_index, 'pos_ids': pos_ids, 'cosine': cosine,
           'mask_pad_pos': mask_pad_pos, 'sparse_grad': sparse_grad, 'n_queries': M}
    kind = _sampler_kind(sampler) if sampler is not None else nat.SAMPLER_GIVEN
    if kind is None:
        raise TypeError(f'fused path does not cover sampler {type(sampler).__name__}')
    cfg['sampler'] = kind
    if kind == nat.SAMPLER_GIVEN:
        if neg_ids is None:
            raise ValueError('neg_ids is required when no sampler is given')
        cfg['neg_ids'] = neg_ids.reshape(M, -1)
    elif kind == nat.SAMPLER_POPULAR:
        cfg.update(sampler.lookup_kwargs())
    pos_score, neg_score = _ScoreFn.apply(item_weight, query_src, cfg)
    out = cfg['out']
    score = {'pos_score': pos_score if pos_ids is not None else None, 'neg_score': neg_score}
    if kind == nat.SAMPLER_POPULAR:
        score['log_pos_prob'] = out.get('pos_logp')
        score['log_neg_prob'] = out['neg_logp']
    else:
        # UniformSampler.compute_item_p: int64 zeros (sampler.py:113-114)
        score['log_pos_prob'] = None if pos_ids is None else ops.zero_logp_like(pos_ids)
        score['log_neg_prob'] = ops.zero_logp_like(out['neg_ids'])
    return score, out['neg_ids']


class _FusedBPRFn(torch.autograd.Function):
    """forward + BPRLoss (or SampledSoftmaxLoss, cfg['loss'] = 'ssm') in ONE kernel launch (loss evaluated in the
    epilogue, d loss/d score kept), backward = one write-only launch.  == BaseRetriever.training_step with BPRLoss /
    SampledSoftmaxLoss (baseretriever.py:399-404, loss_func.py:55-59, :80-90)."""

    @staticmethod
    def forward(ctx, item_weight, query_src, cfg):
        # d loss/d query is accumulated by the forward itself while the negative rows are in registers
        # (inner product, stock dims): the backward launch then only writes item-gradient rows
        fwd_qgrad = ctx.needs_input_grad[1] and item_weight.shape[1] in (32, 64, 128, 256)
        loss = cfg.get('loss', 'bpr')
        out = ops.fused_forward(item_weight, query_src, cfg['num_neg'], query_index=cfg.get('query_index'),
                                pos_ids=cfg['pos_ids'], sampler=cfg['sampler'], neg_ids=cfg.get('neg_ids'),
                                n_queries=cfg['n_queries'], want_logp=False, fused_loss=loss, want_query_grad=fwd_qgrad,
                                pos_logp=cfg.get('pos_logp'), neg_logp=cfg.get('neg_logp'), want_scores=False, **_pop_kw(cfg))
        cfg['out'] = out
        ctx.cfg = cfg
        ctx.fwd_qgrad = fwd_qgrad
        ctx.save_for_backward(item_weight, query_src, out['neg_ids'], out['dpos'], out['dneg'])
        return out['loss']

    @staticmethod
    def backward(ctx, g):
        cfg = ctx.cfg
        item_weight, query_src, neg_ids, dpos, dneg = ctx.saved_tensors
        qi, pos_ids, sparse = cfg.get('query_index'), cfg['pos_ids'], cfg.get('sparse_grad', False)
        need_item, need_q = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        have_q = need_q and ctx.fwd_qgrad
        qtab = torch.zeros_like(query_src) if (need_q and not have_q and qi is not None and not sparse) else None
        item_grad = rows = qgrad = None
        sorted_dense = (need_item and not sparse and (have_q or not need_q) and item_weight.shape[1] in (64, 128, 256))
        if sorted_dense:
            # dense weight.grad without atomics: elements sorted by item id, one read-modify-write per touched row
            item_grad = ops.scatter_rows_sorted(torch.zeros_like(item_weight), query_src, neg_ids, dneg, query_index=qi,
                                                pos_ids=pos_ids, dpos=dpos, upstream=g.reshape(1).contiguous(), pad_row=0)
        elif need_item or (need_q and not have_q):
            item_grad, rows, qgrad = ops.fused_backward(
                item_weight, query_src, neg_ids, dneg, query_index=qi, pos_ids=pos_ids, dpos=dpos,
                upstream=g.reshape(1).contiguous(), dense_item_grad=need_item and not sparse,
                row_item_grad=need_item and sparse, want_query_grad=need_q and not have_q and qtab is None,
                query_table_grad=qtab)
        if have_q:
            qgrad = cfg['out']['query_grad'] * g
        g_item = g_q = None
        if need_item:
            if sparse:
                M, n = neg_ids.shape
                idx = torch.cat([pos_ids.view(M, 1), neg_ids], 1).reshape(1, -1)
                g_item = torch.sparse_coo_tensor(idx, rows, item_weight.shape)
            else:
                g_item = item_grad
        if need_q:
            if qi is None:
                g_q = qgrad
            elif sparse:
                g_q = torch.sparse_coo_tensor(qi.view(1, -1), qgrad, query_src.shape)
            elif have_q:
                g_q = ops.scatter_add_rows(qgrad, qi, query_src.shape[0])     # embedding_dense_backward, row 0 skipped
            else:
                g_q = qtab
        return g_item, g_q, None


def fused_ssm_loss(item_weight, query_src, num_neg, *, query_index=None, pos_ids, sampler=None, neg_ids=None,
                   pos_logp=None, neg_logp=None, sparse_grad=False):
    """SampledSoftmaxLoss (loss_func.py:80-90, one positive per row) through the single-launch fused path:
    sampling, gather, scores, the logsumexp over a query's positive and num_neg negatives, the loss and d loss/d query
    in one kernel (num_neg % 64 == 0, inner product, embed_dim in {32, 64, 128, 256}); the backward only writes the
    item-gradient rows.  With ``neg_ids`` given, ``pos_logp`` / ``neg_logp`` are the sampler's log-probabilities
    (None = 0).  Returns (loss, neg_ids)."""
    return fused_bpr_loss(item_weight, query_src, num_neg, query_index=query_index, pos_ids=pos_ids, sampler=sampler,
                          neg_ids=neg_ids, sparse_grad=sparse_grad, loss='ssm', pos_logp=pos_logp, neg_logp=neg_logp)


def fused_bpr_loss(item_weight, query_src, num_neg, *, query_index=None, pos_ids, sampler=None, neg_ids=None,
                   sparse_grad=False, loss='bpr', pos_logp=None, neg_logp=None):
    """BPR (``loss='ssm'``: sampled softmax) training loss through the single-launch fused path
    (num_neg % 64 == 0).  Returns (loss, neg_ids)."""
    M = query_index.numel() if query_index is not None else query_src.shape[0]
    cfg = {'num_neg': int(num_neg), 'query_index': query_index, 'pos_ids': pos_ids, 'sparse_grad': sparse_grad,
           'n_queries': M, 'loss': loss, 'pos_logp': pos_logp, 'neg_logp': neg_logp}
    kind = _sampler_kind(sampler) if sampler is not None else nat.SAMPLER_GIVEN
    if kind is None:
        raise TypeError(f'fused path does not cover sampler {type(sampler).__name__}')
    cfg['sampler'] = kind
    if kind == nat.SAMPLER_GIVEN:
        cfg['neg_ids'] = neg_ids.reshape(M, -1)
    elif kind == nat.SAMPLER_POPULAR:
        cfg.update(sampler.lookup_kwargs())
    loss = _FusedBPRFn.apply(item_weight, query_src, cfg)
    return loss, cfg['out']['neg_ids']


def train_step_no_autograd(item_weight, query_src, num_neg, loss_kind, *, query_index, pos_ids, sampler=None,
                           neg_ids=None, sparse_grad=True, want_user_grad=True):
    """sample+gather+score -> loss(+dscore) -> scatter-add, three launches, no autograd graph.
    Used by bench.py for the training-step figure; returns (loss, score dict, grads)."""
    with torch.no_grad():
        score, ids = retriever_scores(item_weight, query_src, num_neg, query_index=query_index, pos_ids=pos_ids,
                                      sampler=sampler, neg_ids=neg_ids)
        lp = score['log_pos_prob'] if score['log_pos_prob'] is not None and score['log_pos_prob'].is_floating_point() else None
        ln = score['log_neg_prob'] if score['log_neg_prob'].is_floating_point() else None
        loss, dpos, dneg, _ = ops.pairwise_loss(loss_kind, score['pos_score'], score['neg_score'], lp, ln)
        grads = ops.fused_backward(item_weight, query_src, ids, dneg, query_index=query_index, pos_ids=pos_ids,
                                   dpos=dpos, dense_item_grad=not sparse_grad, row_item_grad=sparse_grad,
                                   want_query_grad=want_user_grad)
    return loss, score, grads


def bpr_sgd_step(item_weight, user_weight, num_neg, lr, *, user_ids, pos_ids, sampler=None, neg_ids=None, atomics=False,
                 in_forward=None):
    """One complete SGD training step of a BPR two-tower model (nn.Embedding user and item tables) in three
    launches and WITHOUT gradient tensors: the forward samples, scores, evaluates BPRLoss and accumulates the
    user-row gradients; ``rsa_rows_update_sorted`` adds ``-lr * dneg * q`` straight into the touched ITEM rows of the
    weight table (elements sorted by item id, one read-modify-write per row, bit-reproducible; ``atomics=True``: the
    backward kernel's float atomics instead) and a row scatter applies ``-lr * q.grad`` to the touched USER
    rows.  Equal to ``loss.backward(); torch.optim.SGD(lr).step()`` on the dense gradients (no momentum /
    weight decay) up to fp32 summation order -- without the [N, d] gradient zero-fill, scatter and dense update
    that dominate that path (5.12 GB each at N = 1e7).  Returns (loss, neg_ids).  num_neg % 64 == 0.

    ``in_forward`` (default: on for num_neg == 64 and embed_dim in {64, 128, 256}): most item rows of a step are touched
    by exactly ONE of its elements -- such a row is read by one wave of the forward and by nobody else, so that wave
    rewrites it on the spot (row + (-lr) * d * q, the row and the query fragment being in registers): no second read of
    the row, no read of the query row.  The negatives are drawn first (the Sampler plugin: same ids, same generator
    consumption as the in-kernel sampler) and sorted by item id BEFORE the forward (``rsa_sort_step_elements``: the sort the
    all-sorted form runs after it); a run of length one in the sorted order is a solo row, and only the other elements go
    through the apply pass (``rsa_rows_update_presorted``).  Same result as the all-sorted
    form bit for bit on the solo rows, equal up to fp32 summation order on the shared ones; bit-reproducible."""
    M = user_ids.numel()
    kind = _sampler_kind(sampler) if sampler is not None else nat.SAMPLER_GIVEN
    if kind is None:
        raise TypeError(f'fused path does not cover sampler {type(sampler).__name__}')
    kw = {}
    if kind == nat.SAMPLER_GIVEN:
        kw['neg_ids'] = neg_ids.reshape(M, -1)
    elif kind == nat.SAMPLER_POPULAR:
        kw.update(sampler.lookup_kwargs())
    if in_forward is None:
        in_forward = num_neg == 64 and item_weight.shape[1] in (64, 128, 256) and not atomics
    if in_forward:
        return _bpr_sgd_step_in_forward(item_weight, user_weight, num_neg, lr, user_ids, pos_ids, sampler, kind, kw)
    with torch.no_grad():
        iw, uw = item_weight.data, user_weight.data
        out = ops.fused_forward(iw, uw, num_neg, query_index=user_ids, pos_ids=pos_ids, sampler=kind, want_logp=False,
                                fused_bpr=True, want_query_grad=True, **kw)
        step = torch.full((1,), -float(lr), dtype=torch.float32, device=iw.device)
        # the user-row gradient is complete BEFORE any weight changes (it was accumulated by the forward); the
        # item-row update needs the pre-update user rows, so it runs first
        if iw.shape[1] in (64, 128, 256) and not atomics:
            # sorted by item id: every touched row is read-modified-written once, in a fixed order
            ops.scatter_rows_sorted(iw, uw, out['neg_ids'], out['dneg'], query_index=user_ids, pos_ids=pos_ids,
                                    dpos=out['dpos'], upstream=step, pad_row=0)
        else:
            ops.fused_backward(iw, uw, out['neg_ids'], out['dneg'], query_index=user_ids, pos_ids=pos_ids, dpos=out['dpos'],
                               upstream=step, dense_item_grad=True, item_grad_out=iw, want_query_grad=False)
        _apply_user_rows(uw, user_ids, out['query_grad'], step)
    return out['loss'], out['neg_ids']


def _apply_user_rows(uw, user_ids, qgrad, step):
    """user[uid] += step * qgrad: duplicates of a user in the batch summed in sorted order, no atomics (bit-reproducible)
    for the stock dims; the float-atomic row scatter otherwise."""
    M = user_ids.numel()
    if uw.shape[1] in (64, 128, 256):
        ones = torch.ones(M, 1, dtype=torch.float32, device=uw.device)
        ops.scatter_rows_sorted(uw, qgrad, user_ids.view(M, 1), ones, upstream=step, pad_row=0)
    else:
        ops.scatter_add_rows(qgrad * step, user_ids, uw.shape[0], out=uw)


def _sgd_step_block(iw, uw, num_neg, M, kind, sampler, step_scale, neg=None):
    """The buffers and the frozen ``rsa_bpr_sgd_args`` block of an in-place SGD step of M queries: ONE allocation (``ops.carve``)
    for what the two calls write, the item-AND-user sort workspace (ABI 11), the block filled but for the batch pointers and the
    Philox state."""
    lib, dev, n, d = nat.lib(), iw.device, int(num_neg), iw.shape[1]
    N, U = iw.shape[0], uw.shape[0]
    ws = int(lib.rsa_scatter_rows_sorted_workspace_bytes(M, n + 1, N))       # M * (n + 2) elements: items, positives, users
    f32, u8 = torch.float32, torch.uint8
    specs = [('solo', (M, n + 1), u8), ('iws', (max(ws, 8),), u8), ('pos_score', (M,), f32), ('neg_score', (M, n), f32),
             ('row_loss', (M,), f32), ('dpos', (M,), f32), ('dneg', (M, n), f32), ('query_grad', (M, d), f32)]
    if neg is None:
        specs.insert(0, ('neg', (M, n), torch.int64))
    b = ops.carve(dev, specs)
    if neg is not None:
        b['neg'] = neg
    b['ones'] = torch.ones(M, dtype=torch.float32, device=dev)
    a = nat.BprSgdArgs()
    a.item_table, a.n_items, a.user_table, a.n_users, a.dim, a.num_neg = ptr(iw), N, ptr(uw), U, d, n
    a.n_queries, a.sampler, a.step_scale = M, kind, ptr(step_scale)
    a.neg_ids, a.solo = ptr(b['neg']), ptr(b['solo'])
    a.item_workspace, a.item_workspace_bytes = ptr(b['iws']), ws
    a.pos_score, a.neg_score, a.row_loss = ptr(b['pos_score']), ptr(b['neg_score']), ptr(b['row_loss'])
    a.dpos, a.dneg, a.query_grad, a.ones = ptr(b['dpos']), ptr(b['dneg']), ptr(b['query_grad']), ptr(b['ones'])
    if kind == nat.SAMPLER_UNIFORM:
        a.uniform_high = int(sampler.num_items) + 1
    b['args'], b['ref'] = a, ctypes.byref(a)
    return b


def _reserve_draw(a, kind, sampler, M, num_neg, dev):
    """the Philox state the Sampler plugin's call would consume for this step's negatives -> the argument block"""
    unroll = 4 if kind == nat.SAMPLER_POPULAR else rng.randint_unroll(1, int(sampler.num_items) + 1)
    pc = rng.reserve(M * num_neg, unroll, dev, None)
    a.seed, a.offset, a.grid_threads, a.elem_base = pc.seed, pc.offset, pc.grid_threads, pc.elem_base


def _bpr_sgd_step_in_forward(item_weight, user_weight, num_neg, lr, user_ids, pos_ids, sampler, kind, kw):
    """The in-forward step as the library's two calls on the current stream (``rsa_bpr_sgd_prepare`` / ``_apply``, rsa_step.hip):
    the negatives drawn inside the first launch of ONE sort that covers the item rows and the user rows, solo classification,
    forward with in-place solo updates, sorted apply of the shared item rows and of the user rows."""
    M = user_ids.numel()
    iw, uw = item_weight.data, user_weight.data
    dev = iw.device
    if not (iw.is_cuda and iw.is_contiguous() and uw.is_contiguous() and uw.device == dev and iw.dtype == uw.dtype == torch.float32):
        raise RuntimeError('bpr_sgd_step: contiguous fp32 tables on one GPU (there is no CPU fallback)')
    if not (user_ids.is_cuda and user_ids.dtype == pos_ids.dtype == torch.int64 and user_ids.device == dev and pos_ids.numel() == M):
        raise TypeError('bpr_sgd_step: int64 user / positive ids on the tables\' device, one positive per query')
    user_ids, pos_ids = user_ids.contiguous(), pos_ids.contiguous()
    neg = None
    if kind == nat.SAMPLER_GIVEN:
        neg = kw['neg_ids']
        if not (neg.dtype == torch.int64 and neg.device == dev and neg.shape == (M, num_neg)):
            raise TypeError('bpr_sgd_step: neg_ids must be int64 [n_queries, num_neg] on the tables\' device')
        neg = neg.contiguous()
    with torch.no_grad(), _device_of(dev):
        step = torch.full((1,), -float(lr), dtype=torch.float32, device=dev)
        b = _sgd_step_block(iw, uw, num_neg, M, kind, sampler, step, neg=neg)
        a = b['args']
        if kind == nat.SAMPLER_POPULAR:
            pk = sampler.lookup_kwargs()
            pk.pop('table_prob', None)
            pop = ops.popular_args(**pk)
            a.pop = ctypes.pointer(pop)
        a.user_ids, a.pos_ids = user_ids.data_ptr(), pos_ids.data_ptr()
        if kind != nat.SAMPLER_GIVEN:
            _reserve_draw(a, kind, sampler, M, num_neg, dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        a.loss_out = loss.data_ptr()
        main = torch.cuda.current_stream(dev)
        a.reduce_scratch = ptr(ops._scratch_for(dev, main.cuda_stream))
        lib = nat.lib()
        rc = lib.rsa_bpr_sgd_prepare(b['ref'], main.cuda_stream)
        if rc != 0:
            nat.check(rc, 'rsa_bpr_sgd_prepare')
        rc = lib.rsa_bpr_sgd_apply(b['ref'], main.cuda_stream)
        if rc != 0:
            nat.check(rc, 'rsa_bpr_sgd_apply')
    return loss, b['neg']


class PrefetchedBPRSGD:
    """``bpr_sgd_step`` (in-forward form) with the part of a step that does not depend on the weights -- drawing the
    negatives, sorting the step's (item id, element) pairs, classifying the solo rows -- issued on a SIDE stream, so
    that it runs under the previous step's forward and apply passes instead of in front of its own (those kernels are
    request- and latency-bound: a fifth of the step when they run alone).

        stepper = PrefetchedBPRSGD(item.weight, user.weight, num_neg=64, lr=0.05, sampler=sampler)
        ticket = stepper.prepare(uid0, pos0)
        for uid1, pos1 in following_batches:
            nxt = stepper.prepare(uid1, pos1)          # side stream: overlaps the step below
            loss, neg_ids = stepper.step(ticket)
            ticket = nxt

    ``prepare`` calls consume the sampler's generator in call order, so the negatives -- and therefore every weight -- are
    those of the same sequence of ``bpr_sgd_step`` calls, bit for bit (the work is the same, only its place in time
    moves).  A ticket is stepped exactly once, in the order the tickets were prepared."""

    RING = 2          # buffer sets in flight: the batch being stepped and the one being prepared

    def __init__(self, item_weight, user_weight, num_neg, lr, sampler):
        self.iw, self.uw = item_weight.data, user_weight.data
        if not (num_neg == 64 and self.iw.shape[1] in (64, 128, 256)):
            raise NotImplementedError('PrefetchedBPRSGD: num_neg == 64 and embed_dim in {64, 128, 256} (the in-forward update)')
        self.kind = _sampler_kind(sampler)
        if self.kind not in (nat.SAMPLER_UNIFORM, nat.SAMPLER_POPULAR):
            raise TypeError(f'PrefetchedBPRSGD does not cover sampler {type(sampler).__name__}')
        if not (self.iw.is_cuda and self.iw.is_contiguous() and self.uw.is_contiguous() and self.uw.device == self.iw.device
                and self.iw.dtype == self.uw.dtype == torch.float32):
            raise RuntimeError('PrefetchedBPRSGD: contiguous fp32 tables on one GPU (there is no CPU fallback)')
        self.num_neg, self.sampler, self.dev = int(num_neg), sampler, self.iw.device
        self.side = torch.cuda.Stream(device=self.dev)
        self.step_scale = torch.full((1,), -float(lr), dtype=torch.float32, device=self.dev)
        self._fork = torch.cuda.Event()           # "everything the main stream was given so far", for the side stream to wait on
        self._slots, self._count = {}, 0
        self._lib = nat.lib()
        self._pop = self._pop_src = None

    def set_lr(self, lr):
        """New learning rate from the next ``step`` on (call between steps, e.g. a scheduler at the end of an epoch)."""
        self.step_scale.fill_(-float(lr))

    # -- the argument blocks: one per buffer set and batch size, filled once; a step rewrites the batch pointers and the Philox state
    def _popular_tables(self):
        src = self.sampler._buffers.get('table')
        if self._pop is None or src is not self._pop_src:        # (re)built tables after load_state_dict / .to()
            kw = self.sampler.lookup_kwargs()
            kw.pop('table_prob', None)
            self._pop = ops.popular_args(**kw)
            self._pop_keep = kw
            self._pop_src = src
            for slots in self._slots.values():
                for sl in slots:
                    sl['args'].pop = ctypes.pointer(self._pop)
        return self._pop

    def _slot(self, M):
        slots = self._slots.get(M)
        if slots is None:
            slots = []
            for _ in range(self.RING):
                b = _sgd_step_block(self.iw, self.uw, self.num_neg, M, self.kind, self.sampler, self.step_scale)
                b['ready'] = torch.cuda.Event()
                if self.kind == nat.SAMPLER_POPULAR:
                    b['args'].pop = ctypes.pointer(self._popular_tables())
                slots.append(b)
            self._slots[M] = slots
        return slots[self._count % self.RING]

    def prepare(self, user_ids, pos_ids):
        """Issue the weight-independent part of a step (negatives, sorts, classification) on the side stream -> ticket."""
        if not (user_ids.is_cuda and user_ids.dtype == pos_ids.dtype == torch.int64 and user_ids.is_contiguous()
                and pos_ids.is_contiguous() and user_ids.numel() == pos_ids.numel() and user_ids.device == self.dev):
            raise TypeError('PrefetchedBPRSGD.prepare: contiguous int64 user / positive ids on the tables\' device')
        M = user_ids.numel()
        if self.kind == nat.SAMPLER_POPULAR:
            self._popular_tables()
        b = self._slot(M)
        self._count += 1
        a = b['args']
        a.user_ids, a.pos_ids = user_ids.data_ptr(), pos_ids.data_ptr()
        _reserve_draw(a, self.kind, self.sampler, M, self.num_neg, self.dev)
        main = torch.cuda.current_stream(self.dev)
        self._fork.record(main)                      # the batch tensors may have been produced on the main stream, and the
        self.side.wait_event(self._fork)             # buffer set's previous step must be over
        with _device_of(self.dev):
            rc = self._lib.rsa_bpr_sgd_prepare(b['ref'], self.side.cuda_stream)
        if rc != 0:
            nat.check(rc, 'rsa_bpr_sgd_prepare')
        b['ready'].record(self.side)
        return {'user_ids': user_ids, 'pos_ids': pos_ids, 'neg': b['neg'], 'slot': b, 'seq': self._count}

    def step(self, ticket):
        """The weight-dependent part of the step on the current stream; returns (loss, neg_ids).  ``neg_ids`` (and the
        other per-step buffers) belong to the stepper and are overwritten by the RING-th ``prepare`` call after this
        ticket's: copy what must outlive that."""
        b = ticket['slot']
        if ticket['seq'] + self.RING <= self._count:
            raise RuntimeError('PrefetchedBPRSGD.step: this ticket\'s buffers were reused (tickets are stepped in order, at most '
                               f'{self.RING - 1} prepared ahead)')
        main = torch.cuda.current_stream(self.dev)
        main.wait_event(b['ready'])
        loss = torch.empty((), dtype=torch.float32, device=self.dev)
        a = b['args']
        a.loss_out = loss.data_ptr()
        with _device_of(self.dev):
            a.reduce_scratch = ptr(ops._scratch_for(self.dev, main.cuda_stream))
            rc = self._lib.rsa_bpr_sgd_apply(b['ref'], main.cuda_stream)
        if rc != 0:
            nat.check(rc, 'rsa_bpr_sgd_apply')
        return loss, ticket['neg']


class _device_of:
    """``with _device_of(dev):`` -- dev current for the duration of a native call (a no-op when it already is)."""
    __slots__ = ('dev', 'guard')

    def __init__(self, dev):
        self.dev, self.guard = dev, None

    def __enter__(self):
        if self.dev.index != torch.cuda.current_device():
            self.guard = torch.cuda.device(self.dev)
            self.guard.__enter__()

    def __exit__(self, *exc):
        if self.guard is not None:
            self.guard.__exit__(*exc)


class FusedBPRAdam:
    """Complete lazy-Adam training step of a BPR two-tower model (nn.Embedding user and item tables) without
    gradient tensors: the forward samples, scores, evaluates BPRLoss and accumulates the user-row gradients;
    ``rsa_rows_update_sorted`` sorts the step's (item id, element) pairs, sums every touched row's gradient in
    registers and applies torch.optim.SparseAdam's update to that row of (weight, exp_avg, exp_avg_sq); the user
    rows go through the same kernel with the accumulated row gradients.  Equal (up to fp32 summation order) to
    ``loss.backward()`` with sparse embeddings + ``torch.optim.SparseAdam.step()`` -- whose coalesce pass alone
    takes 16 ms at B = 65 536, n = 64 -- and untouched rows keep their state (lazy)."""

    def __init__(self, item_weight, user_weight, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.iw, self.uw = item_weight.data, user_weight.data
        if self.iw.shape[1] not in (64, 128, 256):
            raise NotImplementedError('FusedBPRAdam: embed_dim must be 64, 128 or 256')
        self.lr, self.betas, self.eps, self.t = float(lr), betas, float(eps), 0
        self.state = {k: torch.zeros_like(w) for k, w in (('im', self.iw), ('iv', self.iw), ('um', self.uw), ('uv', self.uw))}

    def step(self, num_neg, *, user_ids, pos_ids, sampler=None, neg_ids=None):
        """One step; returns (loss, neg_ids).  num_neg % 64 == 0."""
        M = user_ids.numel()
        kind = _sampler_kind(sampler) if sampler is not None else nat.SAMPLER_GIVEN
        if kind is None:
            raise TypeError(f'fused path does not cover sampler {type(sampler).__name__}')
        kw = {}
        if kind == nat.SAMPLER_GIVEN:
            kw['neg_ids'] = neg_ids.reshape(M, -1)
        elif kind == nat.SAMPLER_POPULAR:
            kw.update(sampler.lookup_kwargs())
        self.t += 1
        st = self.state
        with torch.no_grad():
            out = ops.fused_forward(self.iw, self.uw, num_neg, query_index=user_ids, pos_ids=pos_ids, sampler=kind,
                                    want_logp=False, fused_bpr=True, want_query_grad=True, **kw)
            hp = dict(lr=self.lr, betas=self.betas, eps=self.eps, step=self.t)
            # item rows first: their gradient needs the pre-update user rows (the user gradient is already complete)
            ops.adam_rows_sorted(self.iw, st['im'], st['iv'], self.uw, out['neg_ids'], out['dneg'], query_index=user_ids,
                                 pos_ids=pos_ids, dpos=out['dpos'], pad_row=0, **hp)
            ones = torch.ones(M, 1, dtype=torch.float32, device=self.iw.device)
            ops.adam_rows_sorted(self.uw, st['um'], st['uv'], out['query_grad'], user_ids.view(M, 1), ones, pad_row=0, **hp)
        return out['loss'], out['neg_ids']

    # ---- one batch ahead (see PrefetchedBPRSGD): sampling and the two sorts (item rows, user rows) do not read the weights
    def prepare(self, num_neg, *, user_ids, pos_ids, sampler):
        """Draw the negatives of a batch and sort its (item id, element) pairs on a side stream -> ticket for
        ``step_prepared``.  Tickets are stepped once each, in the order they were prepared; the results are those of the
        same sequence of ``step`` calls bit for bit."""
        if _sampler_kind(sampler) not in (nat.SAMPLER_UNIFORM, nat.SAMPLER_POPULAR):
            raise TypeError(f'FusedBPRAdam.prepare does not cover sampler {type(sampler).__name__}')
        if getattr(self, 'side', None) is None:
            self.side = torch.cuda.Stream(device=self.iw.device)
        main = torch.cuda.current_stream(self.iw.device)
        self.side.wait_stream(main)
        with torch.no_grad(), torch.cuda.stream(self.side):
            neg = sampler(torch.empty(user_ids.numel(), 1, device=self.iw.device), num_neg, None)[0]
            _, ws = ops.sort_step_elements(pos_ids, neg, self.iw.shape[0], pad_row=0, want_solo=False)
            _, ws_user = ops.sort_step_elements(None, user_ids.view(-1, 1), self.uw.shape[0], pad_row=0, want_solo=False)
            ready = torch.cuda.Event()
            ready.record(self.side)
        for t in (neg, ws, ws_user):
            t.record_stream(main)
        return {'user_ids': user_ids, 'pos_ids': pos_ids, 'neg': neg, 'ws': ws, 'ws_user': ws_user, 'ready': ready,
                'num_neg': int(num_neg)}

    def step_prepared(self, ticket):
        torch.cuda.current_stream(self.iw.device).wait_event(ticket['ready'])
        uid, pos, neg, n = ticket['user_ids'], ticket['pos_ids'], ticket['neg'], ticket['num_neg']
        M = uid.numel()
        self.t += 1
        st = self.state
        with torch.no_grad():
            out = ops.fused_forward(self.iw, self.uw, n, query_index=uid, pos_ids=pos, neg_ids=neg, sampler=nat.SAMPLER_GIVEN,
                                    want_logp=False, fused_bpr=True, want_query_grad=True)
            hp = dict(lr=self.lr, betas=self.betas, eps=self.eps, step=self.t)
            ops.adam_rows_presorted(self.iw, st['im'], st['iv'], self.uw, ticket['ws'], M, n, out['dneg'], query_index=uid,
                                    dpos=out['dpos'], pad_row=0, **hp)
            ones = torch.ones(M, 1, dtype=torch.float32, device=self.iw.device)
            ops.adam_rows_presorted(self.uw, st['um'], st['uv'], out['query_grad'], ticket['ws_user'], M, 1, ones, pad_row=0, **hp)
        return out['loss'], out['neg_ids']
