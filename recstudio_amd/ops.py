"""Tensor-level wrappers over the C ABI (include/recstudio_amd.h).

PyTorch is used for device memory, streams and the generator state only; every
computation below happens in the HIP kernels of librecstudio_amd.so.  All
tensors must live on a ROCm device -- there is no CPU path.
"""
import ctypes
import functools

import torch

from . import _native as nat
from . import placement, rng
from ._native import ptr


# Integer all-zero log-probabilities (what the reference's UniformSampler hands out, sampler.py:113-114) come from a
# small cache of persistent zero buffers (one per element count and device) and are recognised by address, not by
# reading them back: the losses would otherwise need a host sync per step to learn that there is nothing to subtract
# (and the sampler a memset per step).  They are constants: nothing in the package or in the reference writes to them.
_ZERO_LOGP = {}


def zero_logp_like(ids):
    """An int64 all-zero tensor shaped like ``ids`` (a view of a cached constant buffer)."""
    key = (ids.numel(), ids.device)
    z = _ZERO_LOGP.get(key)
    if z is None:
        if len(_ZERO_LOGP) >= 16:              # a handful of batch shapes at most; do not grow without bound
            _ZERO_LOGP.clear()
        z = _ZERO_LOGP[key] = torch.zeros(ids.numel(), dtype=torch.int64, device=ids.device)
    return z.view(ids.shape)


def zero_logp(shape, device):
    """The same constant for a shape / device instead of a template tensor."""
    shape = tuple(int(x) for x in shape)
    numel = 1
    for x in shape:
        numel *= x
    device = torch.device(device)
    if device.type == 'cuda' and device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    key = (numel, device)
    z = _ZERO_LOGP.get(key)
    if z is None:
        if len(_ZERO_LOGP) >= 16:
            _ZERO_LOGP.clear()
        z = _ZERO_LOGP[key] = torch.zeros(numel, dtype=torch.int64, device=device)
    return z.view(shape)


def is_known_zero(t):
    if t.is_floating_point():
        return False
    z = _ZERO_LOGP.get((t.numel(), t.device))
    return z is not None and z.data_ptr() == t.data_ptr()


# Every native launch runs on the CURRENT STREAM OF THE DEVICE ITS TENSORS LIVE ON, with that device made current
# for the duration of the call (torch.cuda.device guard): a model on cuda:1 while cuda:0 is the process's current
# device (BaseRetriever honours train.gpu = [1] without calling set_device) must not launch on device 0's stream
# against device-1 pointers.  All tensor arguments of one call must share the device.
_LAUNCH_DEV = []


def _raw_stream(dev=None):
    """hipStream_t (as an int) of the current stream of ``dev``.  torch.cuda.current_stream() builds a Stream object
    (6 us a call, seven calls per sharded step); the raw getter is a plain C call."""
    index = dev.index if dev is not None and dev.index is not None else torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(index)


def _stream():
    return ctypes.c_void_p(_raw_stream(_LAUNCH_DEV[-1] if _LAUNCH_DEV else None))


def _tensor_device(args, kwargs):
    dev = None
    for a in list(args) + list(kwargs.values()):
        if isinstance(a, torch.Tensor) and a.is_cuda:
            if dev is None:
                dev = a.device
            elif a.device != dev:
                raise RuntimeError(f'recstudio_amd: tensors of one call live on different devices ({dev} and {a.device})')
    return dev


def _on_device(fn):
    """Run ``fn`` under the device guard of its tensor arguments (see above)."""
    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        dev = _tensor_device(args, kwargs)
        if dev is None:
            d = kwargs.get('device')
            if d is None:
                d = next((a for a in args if isinstance(a, (torch.device, str))), None)
            dev = torch.device(d) if d is not None else None
            if dev is not None and dev.type == 'cuda' and dev.index is None:
                dev = torch.device('cuda', torch.cuda.current_device())
        if dev is None or dev.type != 'cuda':
            return fn(*args, **kwargs)       # CPU tensors: the op's own _need() raises the "GPU only" error
        _LAUNCH_DEV.append(dev)
        try:
            if dev.index == torch.cuda.current_device():
                return fn(*args, **kwargs)
            with torch.cuda.device(dev):
                return fn(*args, **kwargs)
        finally:
            _LAUNCH_DEV.pop()
    return wrapped


# Caller-owned reduction scratch of the C ABI (rsa_scratch_bytes(): arrival counter + partial sums of the in-kernel
# loss reduction, rsa_mean_rows' partials, the BCE valid-row counter): one zero-filled block per (device, stream),
# so that launches on different streams or devices never share words.
_SCRATCH = {}


def _scratch():
    dev = _LAUNCH_DEV[-1] if _LAUNCH_DEV else torch.device('cuda', torch.cuda.current_device())
    return _scratch_for(dev, _raw_stream(dev))


def _scratch_for(dev, raw_stream):
    key = (dev.index, raw_stream)
    t = _SCRATCH.get(key)
    if t is None:
        if len(_SCRATCH) >= 64:          # short-lived streams: forget the oldest blocks (a block is only ever used on its stream)
            for old in list(_SCRATCH)[:32]:
                del _SCRATCH[old]
        t = _SCRATCH[key] = torch.zeros(int(nat.lib().rsa_scratch_bytes()), dtype=torch.uint8, device=dev)
    return t


def _need(t, dtype, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'{name}: expected a tensor, got {type(t)}')
    if not t.is_cuda:
        raise RuntimeError(f'{name}: recstudio_amd runs on the GPU only (got a {t.device} tensor); '
                           'there is no CPU fallback')
    if t.dtype != dtype:
        raise TypeError(f'{name}: expected {dtype}, got {t.dtype}')
    return t if t.is_contiguous() else t.contiguous()


def _need_opt(t, dtype, name):
    return None if t is None else _need(t, dtype, name)


# ------------------------------------------------------------------ samplers
@_on_device
def sample_uniform(numel, low, high, device, generator=None):
    """int64 [numel], == torch.randint(low, high, (numel,), device=device) incl. generator advance."""
    out = torch.empty(int(numel), dtype=torch.int64, device=device)
    if numel == 0:
        return out
    pc = rng.reserve(numel, rng.randint_unroll(low, high), device, generator)
    nat.check(nat.lib().rsa_sample_uniform(ptr(out), int(numel), int(low), int(high), pc.seed, pc.offset,
                                           pc.grid_threads, pc.elem_base, _stream()), 'rsa_sample_uniform')
    return out


@_on_device
def sample_masked_uniform(user_hist, num_items, per_row, generator=None):
    """int64 [B, per_row]: uniform over [1, num_items] minus each row's history (sampler.py:117-147)."""
    user_hist = _need(user_hist, torch.int64, 'user_hist')
    B, Lh = user_hist.shape
    out = torch.empty(B, int(per_row), dtype=torch.int64, device=user_hist.device)
    if B * per_row:
        pc = rng.reserve(B * per_row, 4, user_hist.device, generator)
        nat.check(nat.lib().rsa_sample_masked_uniform(ptr(user_hist), B, Lh, int(num_items), int(per_row), ptr(out),
                                                      pc.seed, pc.offset, pc.grid_threads, pc.elem_base, _stream()),
                  'rsa_sample_masked_uniform')
    return out


def popular_args(table, pop_prob, guide, guide_log2, cdf_lut=None, cdf_lines=None, lines_log2=0):
    """rsa_popular_args with the sampler's tables filled in (checked device fp32 / int32 tensors)"""
    a = nat.PopularArgs()
    a.table, a.pop_prob, a.guide = ptr(table), ptr(pop_prob), ptr(guide)
    a.n_items, a.guide_log2, a.lines_log2 = table.numel(), int(guide_log2), int(lines_log2)
    a.cdf_lut = ptr(_need_opt(cdf_lut, torch.float32, 'cdf_lut'))
    a.cdf_lines = ptr(_need_opt(cdf_lines, torch.float32, 'cdf_lines'))
    return a


@_on_device
def sample_popular(table, pop_prob, guide, guide_log2, numel, generator=None, want_u=False, cdf_lut=None,
                   cdf_lines=None, lines_log2=0):
    table = _need(table, torch.float32, 'table')
    pop_prob = _need(pop_prob, torch.float32, 'pop_prob')
    guide = _need_opt(guide, torch.int32, 'guide')
    dev = table.device
    ids = torch.empty(int(numel), dtype=torch.int64, device=dev)
    logp = torch.empty(int(numel), dtype=torch.float32, device=dev)
    u = torch.empty(int(numel), dtype=torch.float32, device=dev) if want_u else None
    if numel:
        pc = rng.reserve(numel, 4, dev, generator)
        a = popular_args(table, pop_prob, guide, guide_log2, cdf_lut, cdf_lines, lines_log2)
        a.ids, a.logp, a.u_out, a.numel = ptr(ids), ptr(logp), ptr(u), int(numel)
        a.seed, a.offset, a.grid_threads, a.elem_base = pc.seed, pc.offset, pc.grid_threads, pc.elem_base
        nat.check(nat.lib().rsa_sample_popular(ctypes.byref(a), _stream()), 'rsa_sample_popular')
    return (ids, logp, u) if want_u else (ids, logp)


@_on_device
def popular_lookup(table, pop_prob, guide, guide_log2, u, cdf_lut=None, cdf_lines=None, lines_log2=0):
    table = _need(table, torch.float32, 'table')
    pop_prob = _need(pop_prob, torch.float32, 'pop_prob')
    guide = _need_opt(guide, torch.int32, 'guide')
    u = _need(u, torch.float32, 'u')
    ids = torch.empty(u.numel(), dtype=torch.int64, device=u.device)
    logp = torch.empty(u.numel(), dtype=torch.float32, device=u.device)
    a = popular_args(table, pop_prob, guide, guide_log2, cdf_lut, cdf_lines, lines_log2)
    a.u_in, a.ids, a.logp, a.numel = ptr(u), ptr(ids), ptr(logp), u.numel()
    nat.check(nat.lib().rsa_popular_lookup(ctypes.byref(a), _stream()), 'rsa_popular_lookup')
    return ids.view(u.shape), logp.view(u.shape)


@_on_device
def item_logp(pop_prob, ids):
    pop_prob = _need(pop_prob, torch.float32, 'pop_prob')
    ids = _need(ids, torch.int64, 'ids')
    out = torch.empty(ids.shape, dtype=torch.float32, device=ids.device)
    nat.check(nat.lib().rsa_item_logp(ptr(pop_prob), pop_prob.numel(), ptr(ids), ids.numel(), ptr(out), _stream()),
              'rsa_item_logp')
    return out


# ------------------------------------------------------------------ gathers
@_on_device
def embedding_gather(table, ids):
    table = _need(table, torch.float32, 'table')
    ids = _need(ids, torch.int64, 'ids')
    n_rows, dim = table.shape
    out = torch.empty(*ids.shape, dim, dtype=torch.float32, device=table.device)
    nat.check(nat.lib().rsa_embedding_gather(ptr(table), n_rows, dim, ptr(ids), ids.numel(), ptr(out), _stream()),
              'rsa_embedding_gather')
    return out


@_on_device
def scatter_add_rows(src, ids, n_rows, out=None):
    """dst[ids[i]] += src[i] for ids != 0 (embedding_dense_backward with padding_idx = 0).  ``out``: accumulate
    into an existing [n_rows, dim] tensor instead of a fresh zero one."""
    src = _need(src, torch.float32, 'src')
    ids = _need(ids, torch.int64, 'ids')
    dim = src.shape[-1]
    dst = _need(out, torch.float32, 'out') if out is not None else torch.zeros(n_rows, dim, dtype=torch.float32,
                                                                                device=src.device)
    nat.check(nat.lib().rsa_scatter_add_rows(ptr(src), ptr(ids), ids.numel(), dim, ptr(dst), n_rows, _stream()),
              'rsa_scatter_add_rows')
    return dst


@_on_device
def seg_gather(item_table, flat_item_ids, seg_start, seg_end, max_len, want_rows=True, want_ids=True):
    flat = _need(flat_item_ids, torch.int64, 'flat_item_ids')
    s = _need(seg_start, torch.int64, 'seg_start')
    e = _need(seg_end, torch.int64, 'seg_end')
    dev = flat.device
    B = s.numel()
    tab = _need(item_table, torch.float32, 'item_table') if want_rows else None
    dim = tab.shape[1] if tab is not None else 4
    n_items = tab.shape[0] if tab is not None else 1
    ids = torch.empty(B, max_len, dtype=torch.int64, device=dev) if want_ids else None
    rows = torch.empty(B, max_len, dim, dtype=torch.float32, device=dev) if want_rows else None
    lens = torch.empty(B, dtype=torch.int64, device=dev)
    a = nat.SegGatherArgs()
    a.item_table, a.n_items, a.dim, a.max_len = ptr(tab), n_items, dim, int(max_len)
    a.flat_item_ids, a.n_flat, a.seg_start, a.seg_end, a.n_seg = ptr(flat), flat.numel(), ptr(s), ptr(e), B
    a.out_ids, a.out_rows, a.out_len = ptr(ids), ptr(rows), ptr(lens)
    nat.check(nat.lib().rsa_seg_gather(ctypes.byref(a), _stream()), 'rsa_seg_gather')
    return ids, rows, lens


def _loss_args(kind, pos_score, neg_score, pos_logp, neg_logp, n_rows, num_neg, row_loss, loss_out, dpos, dneg):
    a = nat.LossArgs()
    a.loss_kind, a.n_rows, a.num_neg = int(kind), int(n_rows), int(num_neg)
    a.pos_score, a.neg_score, a.pos_logp, a.neg_logp = ptr(pos_score), ptr(neg_score), ptr(pos_logp), ptr(neg_logp)
    a.row_loss, a.loss_out, a.dpos, a.dneg, a.scratch = ptr(row_loss), ptr(loss_out), ptr(dpos), ptr(dneg), ptr(_scratch())
    return a


def carve(dev, specs, align=4096):
    """``{key: tensor}`` for ``specs = [(key, shape, dtype), ...]``, every tensor a view of ONE allocation.

    Why: the buffers a launch WRITES must not come from separate allocations.  The headline launch (four [B, n] output arrays,
    84 MB of the 2.8 GB it moves) ran 404-409 us on every one of 31 layouts of those arrays inside one allocation (any skew
    between them, any shift, arenas of 86 MB .. 1 GB) and 443-471 us on 5 of 9 sets of separately allocated arrays of the same
    shapes in the same processes (profiles/r05_output_placement.json; moving any ONE of the four into an arena did not help).
    The class belongs to the allocation; with ``RSA_PLACEMENT=1`` (opt-in) ``placement.pick`` probes candidates and returns one of
    the fast class -- by default the arena is one plain ``torch.empty``."""
    offs, total = [], 0
    for _, shape, dtype in specs:
        cnt = 1
        for v in shape:
            cnt *= int(v)
        offs.append((total, cnt))
        total += (cnt * torch.empty((), dtype=dtype).element_size() + align - 1) // align * align
    arena = placement.pick(max(total, align), dev)          # plain torch.empty unless placement was asked for
    out = {}
    for (key, shape, dtype), (off, cnt) in zip(specs, offs):
        nb = cnt * torch.empty((), dtype=dtype).element_size()
        out[key] = arena[off:off + nb].view(dtype).view(tuple(int(v) for v in shape))
    return out


# ------------------------------------------------------------------ fused forward / backward
@_on_device
def fused_forward(item_table, query, num_neg, *, query_index=None, pos_ids=None, neg_ids=None,
                  sampler=nat.SAMPLER_GIVEN, cosine=False, mask_pad_pos=False,
                  table=None, pop_prob=None, guide=None, guide_log2=0, generator=None, n_queries=None,
                  out=None, want_logp=True, table_prob=None, fused_bpr=False, want_mean=True, cdf_lut=None,
                  want_query_grad=False, rng_state=None, cdf_lines=None, lines_log2=0, fused_loss=None,
                  pos_logp=None, neg_logp=None, _plan=None, inplace_update=None, n_batches=1, want_scores=True):
    """One launch of rsa_fused_sample_gather_score.  Returns a dict with
    neg_ids [M,n] int64, neg_score [M,n], pos_score [M] (if pos_ids), and for the
    popularity sampler neg_logp [M,n], pos_logp [M].  ``out``: a dict returned by an earlier
    call with the same shapes, whose buffers are overwritten instead of allocating.
    ``want_logp=False`` skips the log-probability outputs of the popularity sampler (BPR ignores
    them, loss_func.py:55-59).  ``fused_bpr=True`` (needs num_neg % 64 == 0 and pos_ids) evaluates
    BPRLoss in the kernel's epilogue: adds ``loss`` (scalar), ``row_loss [M]``, ``dpos [M]``,
    ``dneg [M, n]`` to the result; with ``want_query_grad`` (inner product, dim in {32, 64, 128, 256}) also
    ``query_grad [M, d]`` = d loss / d query row, accumulated while the rows are in registers.
    ``fused_loss='ssm'`` (same conditions, inner product, dim in {32, 64, 128, 256}): SampledSoftmaxLoss in the
    epilogue instead (one wave per query carries the logsumexp over its num_neg / 64 tiles); with ids given,
    ``pos_logp`` / ``neg_logp`` are the INPUT log-probabilities.  ``cdf_lines`` / ``lines_log2``: the bucket-line
    form of the popularity sampler's inverse CDF (PopularSamplerModel.cdf_lines).
    ``want_scores=False`` (with a loss the kernel fuses): ``neg_score`` is not written and not returned -- a training forward
    needs d loss/d score, not the scores (4 B/triplet less).
    ``n_batches`` = S > 1: a QUEUE of S independent batches consumed by one launch -- ``query_index`` / ``pos_ids`` hold the
    S batches back to back ([S * B]); the outputs are those of S consecutive ``fused_forward`` calls over B queries each,
    concatenated (every batch draws its negatives from its own torch call: the generator is advanced S times), ``loss``
    is the [S] vector of the batches' mean losses.  In-kernel samplers only."""
    item_table = _need(item_table, torch.float32, 'item_table')
    query = _need(query, torch.float32, 'query')
    dev = item_table.device
    n_items, dim = item_table.shape
    if query.dim() != 2 or query.shape[1] != dim:
        raise ValueError(f'query must be [rows, {dim}], got {tuple(query.shape)}')
    query_index = _need_opt(query_index, torch.int64, 'query_index')
    pos_ids = _need_opt(pos_ids, torch.int64, 'pos_ids')
    if n_queries is None:
        n_queries = query_index.numel() if query_index is not None else query.shape[0]
    M, n = int(n_queries), int(num_neg)
    if pos_ids is not None and pos_ids.numel() != M:
        raise ValueError('pos_ids must have one id per query')
    a = nat.FusedArgs()
    if sampler == nat.SAMPLER_GIVEN:
        neg_ids = _need(neg_ids, torch.int64, 'neg_ids')
        if neg_ids.numel() != M * n:
            raise ValueError('neg_ids must be [M, num_neg]')
    else:
        neg_ids = out['neg_ids'] if out is not None else None          # (allocated with the other outputs below)
        unroll = 4 if sampler == nat.SAMPLER_POPULAR else rng.randint_unroll(1, n_items)
        if rng_state is not None:
            # graph-capturable form: (seed, device int64 tensor holding the offset); the caller advances it
            # (rsa_rng_advance) and mirrors the consumption into the torch generator afterwards
            seed, offset_dev = rng_state
            cu, mt = rng.device_props(dev)
            a.seed, a.offset, a.grid_threads = int(seed) & 0xFFFFFFFFFFFFFFFF, 0, rng.grid_threads(M * n, cu, mt)
            a.offset_dev = ptr(offset_dev)
        elif n_batches > 1:
            if M % n_batches:
                raise ValueError('n_batches must divide the number of queries')
            per = M // int(n_batches) * n
            pc = rng.reserve(per, unroll, dev, generator, repeat=int(n_batches))
            if pc.elem_base:
                raise NotImplementedError('fused_forward(n_batches=...) inside rng.sharded_stream')
            a.seed, a.offset, a.grid_threads, a.elem_base = pc.seed, pc.offset, pc.grid_threads, 0
            a.n_batches, a.batch_offset_step = int(n_batches), rng.counter_offset(per, pc.grid_threads, unroll)
            want_mean = False
        else:
            pc = rng.reserve(M * n, unroll, dev, generator)
            a.seed, a.offset, a.grid_threads, a.elem_base = pc.seed, pc.offset, pc.grid_threads, pc.elem_base
    reuse = out is not None        # caller-provided output buffers (same keys/shapes as returned)
    if fused_loss is None and fused_bpr:
        fused_loss = 'bpr'
    score_mode = int(cosine) if not isinstance(cosine, bool) else (nat.SCORE_COS if cosine else nat.SCORE_IP)
    composed = fused_loss == 'bpr' and n != 64 and (dim not in (32, 64, 128, 256) or score_mode != nat.SCORE_IP)
    keep_scores = want_scores or fused_loss not in ('bpr', 'ssm') or composed
    if not reuse:
        # every output of the launch out of ONE allocation (carve): see there for why
        f32 = torch.float32
        specs = [('neg_score', (M, n), f32)] if keep_scores else []
        if sampler != nat.SAMPLER_GIVEN:
            specs.insert(0, ('neg_ids', (M, n), torch.int64))
        if pos_ids is not None:
            specs.append(('pos_score', (M,), f32))
        if sampler == nat.SAMPLER_POPULAR and want_logp:
            specs.append(('neg_logp', (M, n), f32))
            if pos_ids is not None:
                specs.append(('pos_logp', (M,), f32))
        if fused_loss in ('bpr', 'ssm'):
            specs += [('dneg', (M, n), f32), ('loss', (), f32), ('row_loss', (M,), f32), ('dpos', (M,), f32)]
            if want_query_grad:
                specs.append(('query_grad', (M, dim), f32))
        out = carve(dev, specs)
        if sampler != nat.SAMPLER_GIVEN:
            neg_ids = out['neg_ids']
    out['neg_ids'] = neg_ids.view(M, n)
    if sampler == nat.SAMPLER_POPULAR:
        table = _need(table, torch.float32, 'table')
        pop_prob = _need(pop_prob, torch.float32, 'pop_prob')
        guide = _need_opt(guide, torch.int32, 'guide')
    a.item_table, a.n_items, a.dim = ptr(item_table), n_items, dim
    a.score_mode = score_mode
    a.query, a.query_index, a.n_query_rows = ptr(query), ptr(query_index), query.shape[0]
    a.pos_ids, a.n_queries, a.num_neg = ptr(pos_ids), M, n
    a.sampler, a.mask_pad_pos, a.guide_log2 = int(sampler), int(bool(mask_pad_pos)), int(guide_log2)
    a.table, a.pop_prob, a.guide = ptr(table), ptr(pop_prob), ptr(guide)
    a.table_prob = ptr(_need_opt(table_prob, torch.float32, 'table_prob'))
    a.cdf_lut = ptr(_need_opt(cdf_lut, torch.float32, 'cdf_lut'))
    a.cdf_lines, a.lines_log2 = ptr(_need_opt(cdf_lines, torch.float32, 'cdf_lines')), int(lines_log2)
    if sampler == nat.SAMPLER_GIVEN and fused_loss == 'ssm' and (pos_logp is not None or neg_logp is not None):
        # ids given: the log-probabilities are INPUTS of the SampledSoftmax epilogue
        a.neg_logp, a.pos_logp = ptr(_need_opt(neg_logp, torch.float32, 'neg_logp')), ptr(_need_opt(pos_logp, torch.float32, 'pos_logp'))
        a.neg_ids = ptr(neg_ids)
    else:
        a.neg_ids, a.neg_logp, a.pos_logp = ptr(neg_ids), ptr(out.get('neg_logp')), ptr(out.get('pos_logp'))
    a.pos_score, a.neg_score = ptr(out.get('pos_score')), ptr(out['neg_score'] if keep_scores else None)
    if composed:
        # queries longer than one tile are walked by one workgroup (no atomics): that kernel covers the stock dims and
        # the inner product; anything else runs the loss as its own (equally deterministic) launch
        if want_query_grad:
            raise ValueError('want_query_grad needs the inner-product scorer and dim in {32, 64, 128, 256}')
        if n_batches > 1:
            raise ValueError('fused_forward(n_batches=...) with a fused BPR loss needs num_neg == 64, or the inner-product scorer '
                             'and dim in {32, 64, 128, 256} (this configuration is composed of two launches)')
        nat.check(nat.lib().rsa_fused_sample_gather_score(ctypes.byref(a), _stream()), 'rsa_fused_sample_gather_score')
        out['loss'], out['dpos'], out['dneg'], out['row_loss'] = pairwise_loss(nat.LOSS_BPR, out['pos_score'], out['neg_score'])
        return out
    if fused_loss is not None:
        if fused_loss not in ('bpr', 'ssm'):
            raise ValueError("fused_loss must be None, 'bpr' or 'ssm'")
        if 'loss' not in out:
            out['loss'] = torch.empty((), dtype=torch.float32, device=dev)
            out['row_loss'] = torch.empty(M, dtype=torch.float32, device=dev)
            out['dpos'] = torch.empty(M, dtype=torch.float32, device=dev)
            out['dneg'] = torch.empty(M, n, dtype=torch.float32, device=dev)
        a.fused_loss = (nat.LOSS_BPR if fused_loss == 'bpr' else nat.LOSS_SSM) + 1
        a.row_loss, a.loss_out = ptr(out['row_loss']), ptr(out['loss'] if want_mean else None)
        if want_mean:
            a.reduce_scratch = ptr(_scratch())
        a.dpos, a.dneg = ptr(out['dpos']), ptr(out['dneg'])
        if want_query_grad:
            if 'query_grad' not in out:
                out['query_grad'] = torch.empty(M, dim, dtype=torch.float32, device=dev)
            a.query_grad = ptr(out['query_grad'])
        if inplace_update is not None:
            # (solo_flags, scale): rows touched by exactly one element of the step (sort_step_elements) are updated inside
            # the forward: item_table += scale * d * q
            solo_flags, upd_scale = inplace_update
            a.solo_flags, a.upd_scale = ptr(_need(solo_flags, torch.uint8, 'solo_flags')), ptr(_need(upd_scale, torch.float32, 'upd_scale'))
    elif want_query_grad:
        raise ValueError('want_query_grad needs a fused loss (fused_bpr=True / fused_loss=...)')
    if n_batches > 1 and (sampler == nat.SAMPLER_GIVEN or rng_state is not None):
        raise ValueError('fused_forward(n_batches=...) draws its negatives in the kernel from the torch generator')
    nat.check(nat.lib().rsa_fused_sample_gather_score(ctypes.byref(a), _stream()), 'rsa_fused_sample_gather_score')
    if n_batches > 1 and fused_loss is not None:
        _queue_losses(out, int(n_batches))
    if _plan is not None:
        _plan.update(args=a, out=out, device=dev, generator=generator, numel=M * n, n_batches=int(n_batches), keep=(item_table, query, query_index, pos_ids,
                                                                                         neg_ids, table, pop_prob, guide))
        _plan['unroll'] = None if sampler == nat.SAMPLER_GIVEN or rng_state is not None else \
            (4 if sampler == nat.SAMPLER_POPULAR else rng.randint_unroll(1, n_items))
    return out


def _queue_losses(out, n_batches):
    """out['loss'] [S] <- the per-batch means of a queue launch's row losses (the kernel's own mean is per LAUNCH and is off for a
    queue), written IN PLACE into one persistent buffer: ``FusedStep`` hands out the same dict on every call."""
    cur = out.get('loss')
    if cur is None or cur.dim() != 1 or cur.numel() != n_batches:
        out['loss'] = torch.empty(n_batches, dtype=torch.float32, device=out['row_loss'].device)
    torch.mean(out['row_loss'].view(n_batches, -1), dim=1, out=out['loss'])


class FusedStep:
    """A ``fused_forward`` call frozen into its argument block: ``step()`` re-launches the same kernel on the same
    buffers, drawing fresh negatives (the generator is advanced exactly as a ``fused_forward`` call would).  What it
    saves is host time only -- tensor checks, the ctypes struct fill: ~35 us of Python per call, more than a B = 4096
    launch runs on the GPU.  The outputs are the dict of the first call (overwritten in place by every step)."""

    def __init__(self, item_table, query, num_neg, **kw):
        if kw.get('out') is not None or kw.get('rng_state') is not None:
            raise ValueError('FusedStep owns its output buffers and reads the torch generator')
        self._p = {}
        self.out = fused_forward(item_table, query, num_neg, _plan=self._p, **kw)
        if 'args' not in self._p:
            raise ValueError('this configuration is composed of several launches: call fused_forward')
        self._fn = nat.lib().rsa_fused_sample_gather_score
        self._ref = ctypes.byref(self._p['args'])

    def __call__(self):
        p = self._p
        if p['unroll'] is not None:
            nb = p.get('n_batches', 1)
            pc = rng.reserve(p['numel'] // nb, p['unroll'], p['device'], p['generator'], repeat=nb)
            a = p['args']
            a.seed, a.offset, a.grid_threads, a.elem_base = pc.seed, pc.offset, pc.grid_threads, pc.elem_base
        rc = self._fn(self._ref, ctypes.c_void_p(_raw_stream(p['device'])))
        if rc != 0:
            nat.check(rc, 'rsa_fused_sample_gather_score')
        if p.get('n_batches', 1) > 1 and 'row_loss' in self.out:      # a queue's per-batch losses are the host side's
            with torch.cuda.device(p['device']):
                _queue_losses(self.out, p['n_batches'])
        return self.out


@_on_device
def score_packed_keys(item_table, query, keys):
    """Scores of (query row, item row) pairs packed as (qrow << 32) | irow (rsa_shard_route's keys): the fused
    gather+score kernel reading the pairs directly."""
    item_table = _need(item_table, torch.float32, 'item_table')
    query = _need(query, torch.float32, 'query')
    keys = _need(keys, torch.int64, 'keys')
    m = keys.numel()
    out = torch.empty(m, dtype=torch.float32, device=keys.device)
    if m == 0:
        return out
    a = nat.FusedArgs()
    a.item_table, a.n_items, a.dim = ptr(item_table), item_table.shape[0], item_table.shape[1]
    a.query, a.n_query_rows = ptr(query), query.shape[0]
    a.n_queries, a.num_neg, a.sampler = m, 1, nat.SAMPLER_GIVEN
    a.packed_keys, a.neg_score = ptr(keys), ptr(out)
    nat.check(nat.lib().rsa_fused_sample_gather_score(ctypes.byref(a), _stream()), 'rsa_fused_sample_gather_score')
    return out


@_on_device
def pairwise_loss(kind, pos_score, neg_score, pos_logp=None, neg_logp=None, want_grad=True):
    """(loss [scalar tensor], dpos [M] | None, dneg [M,n] | None, row_loss [M])."""
    pos_score = _need(pos_score, torch.float32, 'pos_score')
    neg_score = _need(neg_score, torch.float32, 'neg_score')
    M = pos_score.numel()
    n = neg_score.numel() // max(M, 1)
    if neg_score.numel() != M * n or M == 0 or n == 0:
        raise ValueError(f'pairwise_loss: pos {tuple(pos_score.shape)} vs neg {tuple(neg_score.shape)}')
    pos_logp = _need_opt(pos_logp, torch.float32, 'pos_logp')
    neg_logp = _need_opt(neg_logp, torch.float32, 'neg_logp')
    dev = pos_score.device
    row = torch.empty(M, dtype=torch.float32, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    dpos = torch.empty(pos_score.shape, dtype=torch.float32, device=dev) if want_grad else None
    dneg = torch.empty(neg_score.shape, dtype=torch.float32, device=dev) if want_grad else None
    a = _loss_args(kind, pos_score, neg_score, pos_logp, neg_logp, M, n, row, loss, dpos, dneg)
    nat.check(nat.lib().rsa_pairwise_loss(ctypes.byref(a), _stream()), 'rsa_pairwise_loss')
    return loss, dpos, dneg, row


@_on_device
def pairwise_loss_ex(kind, pos_score, neg_score, pos_logp=None, neg_logp=None, param0=0.0, param1=0.0):
    """rsa_pairwise_loss, kinds WeightedBPR / WeightedBCE / Hinge / NCE / CCL: (loss, dpos, dneg)."""
    pos_score = _need(pos_score, torch.float32, 'pos_score')
    neg_score = _need(neg_score, torch.float32, 'neg_score')
    M = pos_score.numel()
    n = neg_score.numel() // max(M, 1)
    if neg_score.numel() != M * n or M == 0 or n == 0:
        raise ValueError(f'pairwise_loss_ex: pos {tuple(pos_score.shape)} vs neg {tuple(neg_score.shape)}')
    pos_logp = _need_opt(pos_logp, torch.float32, 'pos_logp')
    neg_logp = _need_opt(neg_logp, torch.float32, 'neg_logp')
    dev = pos_score.device
    row = torch.empty(M, dtype=torch.float32, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    dpos, dneg = torch.empty_like(pos_score), torch.empty_like(neg_score)
    if not nat.LOSS_WBPR <= int(kind) <= nat.LOSS_CCL:
        raise ValueError(f'pairwise_loss_ex: unknown loss kind {kind}')
    a = _loss_args(kind, pos_score, neg_score, pos_logp, neg_logp, M, n, row, loss, dpos, dneg)
    a.param0, a.param1 = float(param0), float(param1)
    nat.check(nat.lib().rsa_pairwise_loss(ctypes.byref(a), _stream()), 'rsa_pairwise_loss')
    return loss, dpos, dneg


@_on_device
def ssm_shared_loss(pos_score, neg_score, pos_logp=None, neg_logp=None):
    """rsa_ssm_shared_loss: pos_score [B, L], neg_score [B, n].  Returns (loss, dpos, dneg)."""
    pos_score = _need(pos_score, torch.float32, 'pos_score')
    neg_score = _need(neg_score, torch.float32, 'neg_score')
    pos_logp = _need_opt(pos_logp, torch.float32, 'pos_logp')
    neg_logp = _need_opt(neg_logp, torch.float32, 'neg_logp')
    B, L = pos_score.shape
    n = neg_score.shape[1]
    dev = pos_score.device
    row = torch.empty(B, dtype=torch.float32, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    dpos, dneg = torch.empty_like(pos_score), torch.empty_like(neg_score)
    a = _loss_args(nat.LOSS_SSM, pos_score, neg_score, pos_logp, neg_logp, B, n, row, loss, dpos, dneg)
    a.n_pos = L
    nat.check(nat.lib().rsa_ssm_shared_loss(ctypes.byref(a), _stream()), 'rsa_ssm_shared_loss')
    return loss, dpos, dneg


@_on_device
def fused_backward(item_table, query, neg_ids, dneg, *, query_index=None, pos_ids=None, dpos=None, upstream=None,
                   dense_item_grad=True, row_item_grad=False, want_query_grad=True, query_table_grad=None,
                   item_grad_out=None, query_table_pad_row=0, item_pad_row=0, cosine=False):
    """rsa_fused_backward.  Returns (item_grad [N,d] | None, item_grad_rows [M*(n+1), d] | None,
    query_grad [M,d] | None)."""
    item_table = _need(item_table, torch.float32, 'item_table')
    query = _need(query, torch.float32, 'query')
    neg_ids = _need(neg_ids, torch.int64, 'neg_ids')
    dneg = _need(dneg, torch.float32, 'dneg')
    query_index = _need_opt(query_index, torch.int64, 'query_index')
    pos_ids = _need_opt(pos_ids, torch.int64, 'pos_ids')
    dpos = _need_opt(dpos, torch.float32, 'dpos')
    upstream = _need_opt(upstream, torch.float32, 'upstream')
    dev = item_table.device
    n_items, dim = item_table.shape
    M = query_index.numel() if query_index is not None else query.shape[0]
    n = neg_ids.numel() // M
    if item_grad_out is not None:          # accumulate into a caller-provided (zeroed or running) dense buffer
        item_grad = _need(item_grad_out, torch.float32, 'item_grad_out')
    else:
        item_grad = torch.zeros(n_items, dim, dtype=torch.float32, device=dev) if dense_item_grad else None
    bufs = carve(dev, ([('rows', (M * (n + 1), dim), torch.float32)] if row_item_grad else []) +
                 ([('qgrad', (M, dim), torch.float32)] if want_query_grad else []))
    rows, qgrad = bufs.get('rows'), bufs.get('qgrad')
    if rows is not None and pos_ids is None:
        rows.zero_()
    a = nat.BackwardArgs()
    a.item_table, a.n_items, a.dim, a.num_neg = ptr(item_table), n_items, dim, n
    a.query, a.query_index, a.n_query_rows = ptr(query), ptr(query_index), query.shape[0]
    a.pos_ids, a.neg_ids, a.n_queries = ptr(pos_ids), ptr(neg_ids), M
    a.dpos, a.dneg, a.upstream = ptr(dpos), ptr(dneg), ptr(upstream)
    a.item_grad, a.item_grad_rows, a.query_grad = ptr(item_grad), ptr(rows), ptr(qgrad)
    a.query_table_grad = ptr(_need_opt(query_table_grad, torch.float32, 'query_table_grad'))
    a.query_table_pad_row = int(query_table_pad_row)
    a.item_pad_row = int(item_pad_row)
    a.score_mode = int(cosine) if not isinstance(cosine, bool) else (nat.SCORE_COS if cosine else nat.SCORE_IP)
    nat.check(nat.lib().rsa_fused_backward(ctypes.byref(a), _stream()), 'rsa_fused_backward')
    return item_grad, rows, qgrad


def _rows_update_args(target, query, query_index, n_queries, num_neg, dpos, dneg, upstream, pad_row):
    a = nat.RowsUpdateArgs()
    a.query, a.query_index, a.n_query_rows = ptr(query), ptr(_need_opt(query_index, torch.int64, 'query_index')), query.shape[0]
    a.dim, a.n_queries, a.num_neg = target.shape[1], int(n_queries), int(num_neg)
    a.dpos, a.dneg = ptr(_need_opt(dpos, torch.float32, 'dpos')), ptr(dneg)
    a.upstream = ptr(_need_opt(upstream, torch.float32, 'upstream'))
    a.n_items, a.pad_row, a.target = target.shape[0], int(pad_row), ptr(target)
    return a


def _adam_fields(a, exp_avg, exp_avg_sq, lr, betas, eps, step):
    a.exp_avg, a.exp_avg_sq = ptr(_need(exp_avg, torch.float32, 'exp_avg')), ptr(_need(exp_avg_sq, torch.float32, 'exp_avg_sq'))
    a.lr, a.beta1, a.beta2, a.eps, a.step = float(lr), float(betas[0]), float(betas[1]), float(eps), int(step)


@_on_device
def sort_step_elements(pos_ids, neg_ids, n_items, pad_row=0, want_solo=True):
    """rsa_sort_step_elements: the step's (item id, element) pairs sorted by id in a workspace, and the classification
    ``solo [M, 1 + n]`` uint8 (column 0 = the positive): 1 where no other element of the step touches that item row.
    -> (solo, workspace) for ``fused_forward(inplace_update=...)`` + ``scatter_rows_presorted``."""
    neg_ids = _need(neg_ids, torch.int64, 'neg_ids')
    pos_ids = _need_opt(pos_ids, torch.int64, 'pos_ids')
    M = pos_ids.numel() if pos_ids is not None else neg_ids.shape[0]
    n = neg_ids.numel() // max(M, 1)
    w = n + (1 if pos_ids is not None else 0)
    ws_bytes = int(nat.lib().rsa_scatter_rows_sorted_workspace_bytes(M, n, int(n_items)))
    bufs = carve(neg_ids.device, [('ws', (max(ws_bytes, 8),), torch.uint8)] + ([('solo', (M, w), torch.uint8)] if want_solo else []))
    solo, ws = bufs.get('solo'), bufs['ws']                     # solo None: sort only, nothing flagged
    a = nat.RowsUpdateArgs()
    a.pos_ids, a.neg_ids, a.n_queries, a.num_neg, a.n_items, a.pad_row = ptr(pos_ids), ptr(neg_ids), M, n, int(n_items), int(pad_row)
    a.solo, a.workspace, a.workspace_bytes = ptr(solo), ptr(ws), ws_bytes
    nat.check(nat.lib().rsa_sort_step_elements(ctypes.byref(a), _stream()), 'rsa_sort_step_elements')
    return solo, ws


@_on_device
def scatter_rows_presorted(target, query, workspace, n_queries, num_neg, dneg, *, query_index=None, dpos=None, upstream=None,
                           pad_row=0):
    """rsa_rows_update_presorted: the apply pass of scatter_rows_sorted over the pairs ``sort_step_elements`` left in
    ``workspace`` (elements flagged solo there are skipped: the forward has applied them)."""
    target = _need(target, torch.float32, 'target')
    query = _need(query, torch.float32, 'query')
    dneg = _need(dneg, torch.float32, 'dneg')
    n_items, dim = target.shape
    a = _rows_update_args(target, query, query_index, n_queries, num_neg, dpos, dneg, upstream, pad_row)
    a.has_pos, a.workspace, a.workspace_bytes = int(dpos is not None), ptr(workspace), workspace.numel()
    nat.check(nat.lib().rsa_rows_update_presorted(ctypes.byref(a), _stream()), 'rsa_rows_update_presorted')
    return target


@_on_device
def adam_rows_presorted(weight, exp_avg, exp_avg_sq, query, workspace, n_queries, num_neg, dneg, *, lr, betas=(0.9, 0.999),
                        eps=1e-8, step=1, query_index=None, dpos=None, upstream=None, pad_row=0):
    """rsa_rows_update_presorted with the lazy-Adam state: the apply pass of adam_rows_sorted over the pairs ``sort_step_elements(want_solo=False)``
    left in ``workspace``."""
    weight = _need(weight, torch.float32, 'weight')
    query = _need(query, torch.float32, 'query')
    dneg = _need(dneg, torch.float32, 'dneg')
    n_items, dim = weight.shape
    a = _rows_update_args(weight, query, query_index, n_queries, num_neg, dpos, dneg, upstream, pad_row)
    a.has_pos, a.workspace, a.workspace_bytes = int(dpos is not None), ptr(workspace), workspace.numel()
    _adam_fields(a, exp_avg, exp_avg_sq, lr, betas, eps, step)
    nat.check(nat.lib().rsa_rows_update_presorted(ctypes.byref(a), _stream()), 'rsa_rows_update_presorted(adam)')
    return weight


# ------------------------------------------------------------------ full catalog
@_on_device
def row_lse(x, want_softmax=False, scale=1.0):
    x = _need(x, torch.float32, 'x')
    M, n = x.shape
    lse = torch.empty(M, dtype=torch.float32, device=x.device)
    sm = torch.empty_like(x) if want_softmax else None
    nat.check(nat.lib().rsa_row_lse(ptr(x), M, n, ptr(lse), ptr(sm), float(scale), _stream()), 'rsa_row_lse')
    return lse, sm


def _pad_k(item_table, query):
    """The MFMA kernel is built for d in {32, 64, 128}: zero-pad the k dimension of smaller dims (a copy -- only for
    unusual dims; the dot products are unchanged).  d > 128 is handled by the callers (k split into chunks)."""
    dim = item_table.shape[1]
    if dim in (32, 64, 128):
        return item_table, query, dim
    if dim > 128:
        raise NotImplementedError(f'one MFMA pass covers embed_dim <= 128, got {dim} (ops.fullscore splits it)')
    pad = (32 if dim < 32 else 64 if dim < 64 else 128) - dim
    return torch.nn.functional.pad(item_table, (0, pad)), torch.nn.functional.pad(query, (0, pad)), dim + pad


@_on_device
def row_sqnorm(table, score_mode, pad=0):
    """rsa_row_sqnorm: per-row operand of the cosine (1 / ||row||) or Euclidean (||row||^2) full-catalog scores.
    ``pad`` extra (zero) entries are appended (the MFMA kernel reads up to a tile past the last item)."""
    table = _need(table, torch.float32, 'table')
    n, d = table.shape
    out = torch.zeros(n + pad, dtype=torch.float32, device=table.device) if pad else \
        torch.empty(n, dtype=torch.float32, device=table.device)
    nat.check(nat.lib().rsa_row_sqnorm(ptr(table), n, d, int(score_mode), ptr(out), _stream()), 'rsa_row_sqnorm')
    return out


def _score_mode(cosine):
    return int(cosine) if not isinstance(cosine, bool) else (nat.SCORE_COS if cosine else nat.SCORE_IP)


FULLSCORE_MAX_K = 1024


@_on_device
def fullscore(item_table, query, *, want_scores=False, want_lse=False, k=0, items_without_pad=False, score_mode=0):
    """rsa_fullscore: scores of query [B,d] against rows 1.. of item_table [N,d].
    ``items_without_pad``: ``item_table`` is the reference's ``item_vector`` (= weight[1:], no padding
    row); the kernel never touches row 0, so the base pointer is simply moved one row back.
    ``score_mode``: rsa_score_mode (inner product / cosine / Euclidean, scorer.py:5-34).
    Shapes outside the single-pass kernel are composed from it: embed_dim > 128 accumulates materialised scores
    over 128-wide slices of k; top-k beyond 1024 runs on materialised scores (torch.topk)."""
    item_table = _need(item_table, torch.float32, 'item_table')
    query = _need(query, torch.float32, 'query')
    score_mode = _score_mode(score_mode)
    dim0 = item_table.shape[1]
    if dim0 > 128 or k > FULLSCORE_MAX_K:
        return _fullscore_composed(item_table, query, want_scores, want_lse, k, items_without_pad, score_mode)
    dev = item_table.device
    n_items, dim = item_table.shape
    ia = qa = None
    if score_mode != nat.SCORE_IP:       # per-item / per-query operands from the UNPADDED rows
        rows = item_table if items_without_pad else item_table[1:]
        ia = row_sqnorm(rows, score_mode, pad=64)
        qa = row_sqnorm(query, score_mode)
    item_table, query, dim = _pad_k(item_table, query)
    table_ptr = ptr(item_table)
    if items_without_pad:
        n_items += 1
        table_ptr = ctypes.c_void_p(item_table.data_ptr() - dim * 4)
    B = query.shape[0]
    scores = torch.empty(B, n_items - 1, dtype=torch.float32, device=dev) if want_scores else None
    lse = torch.empty(B, dtype=torch.float32, device=dev) if want_lse else None
    tv = torch.empty(B, k, dtype=torch.float32, device=dev) if k else None
    ti = torch.empty(B, k, dtype=torch.int64, device=dev) if k else None
    lib = nat.lib()

    def launch(lo, hi, stream):
        nq = hi - lo
        ws_bytes = int(lib.rsa_fullscore_workspace_bytes(nq, n_items, int(k)))
        ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=dev)
        a = nat.FullscoreArgs()
        a.item_table, a.n_items, a.dim, a.score_mode, a.n_query = table_ptr, n_items, dim, score_mode, nq
        a.query = query.data_ptr() + lo * dim * 4
        a.scores = None if scores is None else scores.data_ptr() + lo * (n_items - 1) * 4
        a.lse = None if lse is None else lse.data_ptr() + lo * 4
        a.topk_val = None if tv is None else tv.data_ptr() + lo * k * 4
        a.topk_idx = None if ti is None else ti.data_ptr() + lo * k * 8
        a.k, a.item_aux, a.workspace, a.workspace_bytes = int(k), ptr(ia), ptr(ws), ws_bytes
        a.query_aux = None if qa is None else qa.data_ptr() + lo * 4
        nat.check(lib.rsa_fullscore(ctypes.byref(a), stream), 'rsa_fullscore')
        return ws

    launch(0, B, _stream())
    return scores, lse, tv, ti


def _fullscore_composed(item_table, query, want_scores, want_lse, k, items_without_pad, score_mode):
    """embed_dim > 128 and / or k > 1024 (the reference has neither limit, scorer.py:16, baseretriever.py:385):
    inner products accumulated over <= 128-wide column slices through the MFMA kernel into one materialised
    [B, N-1] matrix, the cosine / Euclidean operands applied to it, then logsumexp (rsa_row_lse) and top-k
    (rsa_row_topk up to 1024, torch.topk beyond)."""
    rows = item_table if items_without_pad else item_table[1:]
    d = rows.shape[1]
    scores = None
    for c0 in range(0, d, 128):
        part = fullscore(rows[:, c0:c0 + 128].contiguous(), query[:, c0:c0 + 128].contiguous(), want_scores=True,
                         items_without_pad=True)[0]
        scores = part if scores is None else scores.add_(part)
    if score_mode == nat.SCORE_COS:
        scores.mul_(row_sqnorm(rows, score_mode)).mul_(row_sqnorm(query, score_mode).view(-1, 1))
    elif score_mode == nat.SCORE_EUC:
        scores.mul_(2.0).sub_(row_sqnorm(rows, score_mode)).sub_(row_sqnorm(query, score_mode).view(-1, 1))
    lse = row_lse(scores)[0] if want_lse else None
    tv = ti = None
    if k:
        if k <= FULLSCORE_MAX_K:
            tv, ti = row_topk(scores, k)
        else:
            tv, ti = torch.topk(scores, k, dim=-1)
        ti = ti + 1
    return (scores if want_scores else None), lse, tv, ti


@_on_device
def fullscore_softmax(item_table, query, lse, row_scale=None, want_query_grad=False, want_probs=True):
    """rsa_fullscore_softmax: probs[b, i-1] = row_scale[b] * exp(<query_b, item_i> - lse[b]) over rows 1.. of
    ``item_table`` (dims in {32, 64, 128}).  ``want_query_grad``: -> (probs, probs @ item_table[1:]), the product
    accumulated on the matrix cores in the same pass (rsa_fullscore_softmax_dq); with ``want_probs=False`` the [B, N-1]
    matrix is neither allocated nor written (-> (None, query_grad))."""
    item_table = _need(item_table, torch.float32, 'item_table')
    query = _need(query, torch.float32, 'query')
    lse = _need(lse, torch.float32, 'lse')
    d_in = query.shape[1]
    item_table, query, dim = _pad_k(item_table, query)
    n_items = item_table.shape[0]
    B = query.shape[0]
    if not want_probs and not want_query_grad:
        raise ValueError('fullscore_softmax: nothing to compute (want_probs=False needs want_query_grad=True)')
    probs = torch.empty(B, n_items - 1, dtype=torch.float32, device=item_table.device) if want_probs else None
    if want_query_grad:
        qgrad = torch.empty(B, dim, dtype=torch.float32, device=item_table.device)
        ws_bytes = int(nat.lib().rsa_fullscore_softmax_dq_workspace_bytes(B, n_items, dim))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=item_table.device)
        nat.check(nat.lib().rsa_fullscore_softmax_dq(ptr(item_table), n_items, dim, ptr(query), B, ptr(lse),
                                                     ptr(_need_opt(row_scale, torch.float32, 'row_scale')), ptr(probs),
                                                     ptr(qgrad), ptr(ws), ws_bytes, _stream()), 'rsa_fullscore_softmax_dq')
        return probs, (qgrad if dim == d_in else qgrad[:, :d_in].contiguous())
    nat.check(nat.lib().rsa_fullscore_softmax(ptr(item_table), n_items, dim, ptr(query), B, ptr(lse),
                                              ptr(_need_opt(row_scale, torch.float32, 'row_scale')), ptr(probs),
                                              _stream()), 'rsa_fullscore_softmax')
    return probs


@_on_device
def fullscore_lse_grad(item_table, query):
    """rsa_fullscore_lse_grad: (lse [B], d lse/d query [B, d] = softmax @ item_table[1:]) in ONE pass over the catalog (the flash
    forward of the full softmax).  embed_dim <= 128."""
    item_table = _need(item_table, torch.float32, 'item_table')
    query = _need(query, torch.float32, 'query')
    d_in = query.shape[1]
    tab, q, dim = _pad_k(item_table, query)
    n_items, B, dev = tab.shape[0], q.shape[0], tab.device
    lse = torch.empty(B, dtype=torch.float32, device=dev)
    gq = torch.empty(B, dim, dtype=torch.float32, device=dev)
    ws_bytes = int(nat.lib().rsa_fullscore_lse_grad_workspace_bytes(B, n_items, dim))
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    nat.check(nat.lib().rsa_fullscore_lse_grad(ptr(tab), n_items, dim, ptr(q), B, ptr(lse), ptr(gq), ptr(ws), ws_bytes, _stream()),
              'rsa_fullscore_lse_grad')
    return lse, (gq if dim == d_in else gq[:, :d_in].contiguous())


@_on_device
def fullscore_softmax_dw(item_table, query, lse, row_scale=None, out=None):
    """rsa_fullscore_softmax_dw: d/d item_table of sum_b row_scale[b] * logsumexp_i <query_b, item_i> (rows 1.. of the table; row
    0 of the result is zero) with the softmax tile recomputed on the matrix cores -- no [B, N] matrix.  embed_dim <= 128."""
    item_table = _need(item_table, torch.float32, 'item_table')
    query = _need(query, torch.float32, 'query')
    lse = _need(lse, torch.float32, 'lse')
    d_in = item_table.shape[1]
    tab, q, dim = _pad_k(item_table, query)
    n_items, B = tab.shape[0], q.shape[0]
    direct = dim == d_in and out is not None
    res = out if direct else torch.empty(n_items, dim, dtype=torch.float32, device=tab.device)
    res = _need(res, torch.float32, 'out')
    if res.shape != (n_items, dim):
        raise ValueError(f'fullscore_softmax_dw: out must be [{n_items}, {dim}]')
    nat.check(nat.lib().rsa_fullscore_softmax_dw(ptr(tab), n_items, dim, ptr(q), B, ptr(lse),
                                                 ptr(_need_opt(row_scale, torch.float32, 'row_scale')), ptr(res), _stream()),
              'rsa_fullscore_softmax_dw')
    if direct:
        return out
    res = res if dim == d_in else res[:, :d_in]
    if out is not None:
        out.copy_(res)
        return out
    return res.contiguous()


@_on_device
def probs_t_query(probs, query, out=None):
    """rsa_probs_t_query: ``probs.t() @ query`` ([B, n] x [B, d] -> [n, d]) on the fp32 matrix cores, item-stationary -- the
    d/d items GEMM of the full-softmax backward.  d in {32, 64, 128} (other dims: zero-padded columns, sliced off)."""
    probs = _need(probs, torch.float32, 'probs')
    query = _need(query, torch.float32, 'query')
    B, n = probs.shape
    d = query.shape[1]
    if query.shape[0] != B:
        raise ValueError('probs_t_query: probs [B, n] and query [B, d] must share B')
    if d not in (32, 64, 128):
        if d > 128:
            raise NotImplementedError('probs_t_query: embed_dim <= 128')
        dp = 32 if d <= 32 else (64 if d <= 64 else 128)
        qp = torch.zeros(B, dp, dtype=torch.float32, device=query.device)
        qp[:, :d] = query
        res = probs_t_query(probs, qp)[:, :d]
        if out is not None:
            out.copy_(res)
            return out
        return res.contiguous()
    if out is None:
        out = torch.empty(n, d, dtype=torch.float32, device=probs.device)
    elif tuple(out.shape) != (n, d) or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError('probs_t_query: out must be a contiguous fp32 [n, d] tensor')
    nat.check(nat.lib().rsa_probs_t_query(ptr(probs), B, n, probs.stride(0), ptr(query), d, ptr(out), _stream()), 'rsa_probs_t_query')
    return out


@_on_device
def topk_mask_history(cand_val, cand_idx, user_hist, k):
    """baseretriever.py:386-392 on sorted candidates: drop history items, keep the k best."""
    cand_val = _need(cand_val, torch.float32, 'cand_val')
    cand_idx = _need(cand_idx, torch.int64, 'cand_idx')
    user_hist = _need(user_hist, torch.int64, 'user_hist')
    B, kc = cand_val.shape
    out_v = torch.empty(B, k, dtype=torch.float32, device=cand_val.device)
    out_i = torch.empty(B, k, dtype=torch.int64, device=cand_val.device)
    nat.check(nat.lib().rsa_topk_mask_history(ptr(cand_val), ptr(cand_idx), kc, ptr(user_hist), user_hist.shape[1], B,
                                              int(k), ptr(out_v), ptr(out_i), _stream()), 'rsa_topk_mask_history')
    return out_v, out_i


@_on_device
def row_topk(values, k):
    """torch.topk(values, k) over the last dim (k <= 1024) -> (values, column indices)."""
    values = _need(values, torch.float32, 'values')
    lead, n = values.shape[:-1], values.shape[-1]
    v2 = values.reshape(-1, n)
    out_v = torch.empty(v2.shape[0], k, dtype=torch.float32, device=values.device)
    out_i = torch.empty(v2.shape[0], k, dtype=torch.int64, device=values.device)
    nat.check(nat.lib().rsa_row_topk(ptr(v2), v2.shape[0], n, int(k), ptr(out_v), ptr(out_i), _stream()), 'rsa_row_topk')
    return out_v.view(*lead, k), out_i.view(*lead, k)


@_on_device
def rng_advance(offset_dev, increment):
    """*offset_dev += increment on the stream (rsa_rng_advance): the device copy of the generator offset."""
    nat.check(nat.lib().rsa_rng_advance(ptr(offset_dev), int(increment), _stream()), 'rsa_rng_advance')


@_on_device
def scatter_rows_sorted(target, query, neg_ids, dneg, *, query_index=None, pos_ids=None, dpos=None, upstream=None,
                        pad_row=0):
    """rsa_rows_update_sorted: target[id] += upstream * sum_e d_e * query[qrow_e], sorted by id, no atomics.
    ``target``: [n_items, d] (a zeroed dense gradient, or the weight table with upstream = -lr)."""
    target = _need(target, torch.float32, 'target')
    query = _need(query, torch.float32, 'query')
    neg_ids = _need(neg_ids, torch.int64, 'neg_ids')
    dneg = _need(dneg, torch.float32, 'dneg')
    query_index = _need_opt(query_index, torch.int64, 'query_index')
    pos_ids = _need_opt(pos_ids, torch.int64, 'pos_ids')
    dpos = _need_opt(dpos, torch.float32, 'dpos')
    upstream = _need_opt(upstream, torch.float32, 'upstream')
    n_items, dim = target.shape
    M = query_index.numel() if query_index is not None else query.shape[0]
    if M == 0:
        return target
    n = neg_ids.numel() // M
    ws_bytes = int(nat.lib().rsa_scatter_rows_sorted_workspace_bytes(M, n, n_items))
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=target.device)
    a = _rows_update_args(target, query, query_index, M, n, dpos, dneg, upstream, pad_row)
    a.pos_ids, a.neg_ids, a.workspace, a.workspace_bytes = ptr(pos_ids), ptr(neg_ids), ptr(ws), ws_bytes
    nat.check(nat.lib().rsa_rows_update_sorted(ctypes.byref(a), _stream()), 'rsa_rows_update_sorted')
    return target


@_on_device
def adam_rows_sorted(weight, exp_avg, exp_avg_sq, query, neg_ids, dneg, *, lr, betas=(0.9, 0.999), eps=1e-8, step=1,
                     query_index=None, pos_ids=None, dpos=None, upstream=None, pad_row=0):
    """rsa_rows_update_sorted with the lazy-Adam state: lazy Adam (torch.optim.SparseAdam's rule) on the rows touched by the step, from the
    factored gradient (ids, coefficients, query rows) -- no gradient tensor."""
    weight = _need(weight, torch.float32, 'weight')
    exp_avg = _need(exp_avg, torch.float32, 'exp_avg')
    exp_avg_sq = _need(exp_avg_sq, torch.float32, 'exp_avg_sq')
    query = _need(query, torch.float32, 'query')
    neg_ids = _need(neg_ids, torch.int64, 'neg_ids')
    dneg = _need(dneg, torch.float32, 'dneg')
    query_index = _need_opt(query_index, torch.int64, 'query_index')
    pos_ids = _need_opt(pos_ids, torch.int64, 'pos_ids')
    dpos = _need_opt(dpos, torch.float32, 'dpos')
    upstream = _need_opt(upstream, torch.float32, 'upstream')
    n_items, dim = weight.shape
    M = query_index.numel() if query_index is not None else query.shape[0]
    if M == 0:
        return weight
    n = neg_ids.numel() // M
    ws_bytes = int(nat.lib().rsa_scatter_rows_sorted_workspace_bytes(M, n, n_items))
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=weight.device)
    a = _rows_update_args(weight, query, query_index, M, n, dpos, dneg, upstream, pad_row)
    a.pos_ids, a.neg_ids, a.workspace, a.workspace_bytes = ptr(pos_ids), ptr(neg_ids), ptr(ws), ws_bytes
    _adam_fields(a, exp_avg, exp_avg_sq, lr, betas, eps, step)
    nat.check(nat.lib().rsa_rows_update_sorted(ctypes.byref(a), _stream()), 'rsa_rows_update_sorted(adam)')
    return weight
