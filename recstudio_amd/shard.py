"""Row-sharded item table over the GPUs of one node (BASELINE.json configs[3]).

One process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm).  Rank r owns item
rows [r*rows_per_shard, (r+1)*rows_per_shard) -- or, ``RowShardPlan(layout='interleaved')``, rows r, r + G, r + 2G, ... --
and B queries per step.  Per step (``exchange='fixed'``, the default):

  1. all_gather of the [B, d] query block                                      (RCCL all-gather, own communicator)
  2. ONE launch draws the negatives of the own queries (in-kernel Philox / inverse CDF, one job-wide stream) and
     counting-sorts the B*(1+n) (query, item) elements by owning rank into fixed-capacity, self-describing
     segments: {live count, dropped count} header + 8-byte keys (query index << 32 | local row); each element keeps
     ONE int32: the slot of its key == the slot of its score on the way back              (rsa_shard_sample_route)
  3. equal-split all_to_all of the segments                                     (RCCL all-to-all)
  4. local gather + score on the owner against the gathered queries, straight from the received segments; tiles past
     a segment's count are skipped                                               (rsa_shard_score_segments)
  5. equal-split all_to_all of the fp32 scores back                             (RCCL all-to-all)
  6. ONE launch gathers the scores through the slots and evaluates BPR / SampledSoftmax, its mean and d loss/d score
     (written in routed order for the gradient exchange)                         (rsa_shard_home)

12 bytes per triplet cross xGMI instead of a 512-byte row.  The reference has nothing
comparable: its only multi-device mode re-broadcasts every parameter each step
(recstudio/utils/data_parallel.py:106-159) and DDP is dead code (recommender.py:731-740).

No host round trip in the step: the split is equal, the headers say what is live, and the capacity comes from one
calibration launch (exact counts of the very draw the first step routes, max over slices, owners and ranks, plus
slack).  An element that finds its segment full is DROPPED with a defined outcome -- no key, no score, no loss term,
zero gradient -- and counted in the header; the headers of all sources reach every rank with the keys, so every rank
knows the job-wide dropped count of the step and scales that step's GRADIENTS (or in-place SGD updates) by 0 on the
device: with the in-kernel SGD an overflowed step changes no weight anywhere; a torch optimizer then steps on zero
gradients (plain SGD: nothing moves; Adam / momentum: only their decaying history does).  The count is also kept in a sticky device word that ``check_overflow`` reads off the
critical path (same value on every rank, no collective).  ``exchange='exact'`` is the variable-split form (counts
exchanged and read back each step, ids materialised, separate scatter and loss kernels).

G-invariant negatives: all ranks draw from ONE Philox stream (``sample_generator``, same seed everywhere) by global
element index -- rank r owns rows [r*B, (r+1)*B) of a virtual [G*B, n] id tensor (``rng.sharded_stream``) -- so a
run's negatives do not depend on the number of GPUs.

All device work goes through ``backend`` (default: the HIP kernels).  The protocol itself --
split sizes, exchange order, reassembly -- is backend-agnostic so that the world_size-2 ``gloo``
tests can drive it on CPU with a checker backend supplied BY THE TEST; this module contains no
CPU implementation of the compute.
"""
import ctypes
import weakref

import collections
import torch

from . import _native as nat
from . import ops
from . import rng
from ._native import ptr


class RowShardPlan:
    """Which rank owns which row of the [n_items, d] table.

    ``layout='block'`` (default): contiguous row blocks, owner(id) = id // rows_per_shard.
    ``layout='interleaved'``: owner(id) = id % world, row inside the owner's table = id // world -- every rank holds
    1/world of ANY id range, so a catalog whose ids are ordered by popularity (the hot head of a Zipf law in one block)
    does not make a hot shard whose receive count would set every rank's fixed-capacity segments (the capacity is the
    maximum over the owners)."""

    def __init__(self, n_items, world, layout='block'):
        if layout not in ('block', 'interleaved'):
            raise ValueError("layout must be 'block' or 'interleaved'")
        self.n_items, self.world, self.layout = int(n_items), int(world), layout
        self.interleaved = layout == 'interleaved'
        self.rows_per_shard = (self.n_items + self.world - 1) // self.world
        self.rows_arg = 0 if self.interleaved else self.rows_per_shard      # the C ABI's rows_per_shard (0: interleaved)

    def bounds(self, rank):
        """[lo, hi) of a block plan's rank."""
        if self.interleaved:
            raise ValueError('an interleaved plan has no row bounds: use n_local / take / global_ids')
        lo = min(self.n_items, rank * self.rows_per_shard)
        return lo, min(self.n_items, lo + self.rows_per_shard)

    def n_local(self, rank):
        if self.interleaved:
            return max(0, (self.n_items - rank + self.world - 1) // self.world)
        lo, hi = self.bounds(rank)
        return hi - lo

    def take(self, full, rank):
        """Rank ``rank``'s rows of a full [n_items, ...] tensor (a view)."""
        if self.interleaved:
            return full[rank::self.world]
        lo, hi = self.bounds(rank)
        return full[lo:hi]

    def global_ids(self, rank, local):
        """Item ids of rows ``local`` (tensor or int) of rank ``rank``'s table."""
        return local * self.world + rank if self.interleaved else local + self.bounds(rank)[0]

    def owner(self, ids):
        if self.interleaved:
            return torch.clamp(ids, min=0) % self.world
        return torch.clamp(ids // self.rows_per_shard, 0, self.world - 1)

    def local(self, ids):
        """Row of ``ids`` inside their owners' tables."""
        if self.interleaved:
            return torch.clamp(ids, min=0) // self.world
        return torch.clamp(ids - self.owner(ids) * self.rows_per_shard, min=0)

    def assemble(self, parts):
        """The full table from the ranks' tables (rank order)."""
        if not self.interleaved:
            return torch.cat(list(parts))
        parts = list(parts)
        full = parts[0].new_empty((self.n_items,) + tuple(parts[0].shape[1:]))
        for r, t in enumerate(parts):
            full[r::self.world] = t
        return full


class HipBackend:
    """Device work of the sharded step, all through the C ABI."""

    HDR = 2          # RSA_SHARD_HDR: 8-byte header words {live keys, dropped by the source this step} per segment
    BANKS = 8        # segments per (slice, owner), each filled through its own cursor (rsa_shard_route_args.n_banks)

    def make_generator(self, seed, device):
        return torch.Generator(device=device).manual_seed(int(seed))

    def new_state(self, device):
        """Per-table device words: routing cursors (zeroed once, self-resetting), the sticky job-wide dropped count,
        the dropped count of the last step, and the gated update scales {item scale, gate} the backward multiplies by."""
        return {'cursors': torch.zeros((4096 + 33) * 32, dtype=torch.int32, device=device),
                'overflow': torch.zeros(1, dtype=torch.int32, device=device),
                'step_dropped': torch.zeros(1, dtype=torch.int32, device=device),
                'scale': torch.ones(2, dtype=torch.float32, device=device)}

    def sampler_spec(self, sampler):
        """The in-kernel form of a Sampler plugin (drawn inside the routing pass), or None: then the plugin itself is
        called and its ids are routed as given.  Exact types only -- a subclass may override ``forward``."""
        from .sampler import PopularSamplerModel, UniformSampler
        if type(sampler) is UniformSampler:
            return {'kind': nat.SAMPLER_UNIFORM, 'n_items': sampler.num_items + 1}
        if type(sampler) is PopularSamplerModel:
            return {'kind': nat.SAMPLER_POPULAR, 'n_items': sampler.table.numel(), 'tables': sampler.lookup_kwargs()}
        return None

    def sample(self, sampler, n_queries, n, device, pos_ids, shard=None):
        """The stand-alone Sampler plugin: (log_pos_prob, neg_ids, log_neg_prob).  ``shard = (rank, world, generator)``:
        this rank's rows of ONE global draw (rng.sharded_stream)."""
        q = torch.empty(n_queries, 1, device=device)
        if shard is None:
            return sampler(q, n, pos_ids)
        with rng.sharded_stream(*shard):
            return sampler(q, n, pos_ids)

    def flag_read_async(self, flag):
        """Start copying a device word to the host; returns poll() -> None while in flight, else the value."""
        host = torch.empty(1, dtype=torch.int32).pin_memory()
        host.copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return lambda: int(host[0]) if ev.query() else None

    # -- version 2 of the fixed-capacity exchange (rsa_shard_sample_route / _score_segments / _home) -------------------
    @ops._on_device
    def sample_route(self, state, plan, rank, pos, n, chunks, capacity, spec, generator, neg=None, want_ids=False,
                     want_logp=False, count_only=False, banks=1, route_pos=True, group_by_query=False, deterministic=False):
        """One launch: draw (or read) the negatives, route every (query, item) element of all ``chunks`` slices.
        -> dict(send [C*G*banks*stride] int64, slot_of [B*(1+n)] int32, stride, neg_ids / log_neg_prob / log_pos_prob when
        asked for), or the exact per-segment counts [C*G*banks] int32 with ``count_only`` (the generator is not
        advanced: the next call routes the very draw that was counted)."""
        B, dev, G = pos.numel(), pos.device, plan.world
        a = nat.ShardRouteArgs()
        kind = spec['kind'] if spec is not None else nat.SAMPLER_GIVEN
        a.pos_ids, a.n_queries, a.num_neg, a.sampler = ptr(ops._need(pos, torch.int64, 'pos')), B, int(n), int(kind)
        a.n_slices, a.n_shards, a.n_banks, a.rows_per_shard = int(chunks), G, int(banks), plan.rows_arg
        a.query_base, a.capacity = rank * B, int(capacity)
        a.skip_pos = 0 if route_pos else 1      # the owner-side BPR step scores the positives from the gathered ids
        a.n_items = spec['n_items'] if spec is not None else plan.n_items
        out = {}
        keep = []
        if kind == nat.SAMPLER_GIVEN:
            neg = ops._need(neg, torch.int64, 'neg')
            a.neg_ids = ptr(neg)
            out['neg_ids'] = neg.view(B, n)
        elif B * n:
            unroll = 4 if kind == nat.SAMPLER_POPULAR else rng.randint_unroll(1, a.n_items)
            off0 = generator.get_offset() if count_only else None
            with rng.sharded_stream(rank, G, generator):
                pc = rng.reserve(B * n, unroll, dev)
            if count_only:
                generator.set_offset(off0)
            a.seed, a.offset, a.grid_threads, a.elem_base = pc.seed, pc.offset, pc.grid_threads, pc.elem_base
            if group_by_query and not route_pos and not count_only:
                # whole queries per routing workgroup (n divides 1024, aligned grid): every query's elements for an owner
                # are then ONE contiguous run of a segment and the owner needs no sort by query
                ql = int(nat.lib().rsa_shard_route_query_groups(int(n), int(pc.grid_threads), int(pc.elem_base), int(unroll), G))
                a.group_by_query = 1 if ql > 0 else 0
                out['grouped'] = ql > 0
            if kind == nat.SAMPLER_POPULAR:
                t = spec['tables']
                keep = [ops._need(t['table'], torch.float32, 'table'), ops._need(t['pop_prob'], torch.float32, 'pop_prob')]
                a.table, a.pop_prob = ptr(keep[0]), ptr(keep[1])
                a.guide, a.guide_log2 = ptr(t.get('guide')), int(t.get('guide_log2') or 0)
                a.table_prob, a.cdf_lut = ptr(t.get('table_prob')), ptr(t.get('cdf_lut'))
                a.cdf_lines, a.lines_log2 = ptr(t.get('cdf_lines')), int(t.get('lines_log2') or 0)
            if want_ids and not count_only:
                out['neg_ids'] = torch.empty(B, n, dtype=torch.int64, device=dev)
                a.neg_ids = ptr(out['neg_ids'])
        if kind == nat.SAMPLER_POPULAR and want_logp and not count_only:
            out['log_neg_prob'] = torch.empty(B, n, dtype=torch.float32, device=dev)
            out['log_pos_prob'] = torch.empty(B, dtype=torch.float32, device=dev)
            a.neg_logp, a.pos_logp = ptr(out['log_neg_prob']), ptr(out['log_pos_prob'])
        a.cursors = ptr(state['cursors'])
        if not count_only and state.get('lookup_dropped') is not None:
            a.extra_dropped = ptr(state['lookup_dropped'])     # drops of this step's tower look-ups: gated like a routing overflow
        if count_only:
            counts = torch.empty(chunks * G * banks, dtype=torch.int32, device=dev)
            a.counts_out = ptr(counts)
            nat.check(nat.lib().rsa_shard_sample_route(ctypes.byref(a), ops._stream()), 'rsa_shard_sample_route')
            return counts
        stride = int(capacity) + self.HDR
        out['send'] = torch.empty(chunks * G * banks * stride, dtype=torch.int64, device=dev)
        out['slot_of'] = torch.empty(B * (n + 1), dtype=torch.int32, device=dev)
        out['stride'] = stride
        a.send_keys, a.slot_of = ptr(out['send']), ptr(out['slot_of'])
        if deterministic:
            # no atomic decides a slot: count pass + prefix over the workgroups + routing pass (run-to-run bit-equal segments)
            ints = 2 * int(nat.lib().rsa_shard_route_workgroups(ctypes.byref(a))) * G
            wg = state.get('wg_scratch')
            if wg is None or wg.numel() < ints:
                wg = state['wg_scratch'] = torch.empty(max(ints, 1), dtype=torch.int32, device=dev)
            a.deterministic, a.wg_scratch, a.wg_scratch_ints = 1, ptr(wg), wg.numel()
        nat.check(nat.lib().rsa_shard_sample_route(ctypes.byref(a), ops._stream()), 'rsa_shard_sample_route')
        return out

    @ops._on_device
    def score_segments(self, state, item_local, q_all, recv_keys, n_seg, stride, first=True, out=None):
        """Owner side: fp32 scores of the live slots of ``recv_keys`` [n_seg, stride]; ``first``: also publish the
        step's job-wide dropped count (sum of the received headers) into the state words."""
        scores = out if out is not None else torch.empty(n_seg * stride, dtype=torch.float32, device=recv_keys.device)
        nat.check(nat.lib().rsa_shard_score_segments(
            ptr(item_local), item_local.shape[0], item_local.shape[1], ptr(q_all), q_all.shape[0], ptr(recv_keys), n_seg,
            stride, ptr(scores), ptr(state['step_dropped']) if first else None, ptr(state['overflow']) if first else None,
            ops._stream()), 'rsa_shard_score_segments')
        return scores

    @ops._on_device
    def home(self, scores_home, slot_of, B, n, loss=None, pos_logp=None, neg_logp=None, mean_den=None,
             want_scores=True, want_grad=False, want_dsend=False):
        """Home side: gather through the slots + loss + mean (+ gradients) in one launch.  -> dict."""
        dev = scores_home.device
        a = nat.ShardHomeArgs()
        a.scores, a.slot_of, a.n_queries, a.num_neg = ptr(scores_home), ptr(slot_of), int(B), int(n)
        a.loss = {None: 0, 'bpr': 1, 'ssm': 2}[loss]
        out = {}
        if want_scores or loss is None:
            out['pos_score'] = torch.empty(B, dtype=torch.float32, device=dev)
            out['neg_score'] = torch.empty(B, n, dtype=torch.float32, device=dev)
            a.pos_score, a.neg_score = ptr(out['pos_score']), ptr(out['neg_score'])
        if loss is not None:
            a.mean_den = int(mean_den if mean_den is not None else B)
            a.pos_logp, a.neg_logp = ptr(ops._need_opt(pos_logp, torch.float32, 'pos_logp')), ptr(ops._need_opt(neg_logp, torch.float32, 'neg_logp'))
            out['loss'] = torch.empty((), dtype=torch.float32, device=dev)
            out['row_loss'] = torch.empty(B, dtype=torch.float32, device=dev)
            a.loss_out, a.row_loss, a.reduce_scratch = ptr(out['loss']), ptr(out['row_loss']), ptr(ops._scratch())
            if want_grad:
                out['dpos'] = torch.empty(B, dtype=torch.float32, device=dev)
                out['dneg'] = torch.empty(B, n, dtype=torch.float32, device=dev)
                a.dpos, a.dneg = ptr(out['dpos']), ptr(out['dneg'])
            if want_dsend:
                out['d_send'] = torch.empty(scores_home.numel(), dtype=torch.float32, device=dev)
                a.d_send = ptr(out['d_send'])
        nat.check(nat.lib().rsa_shard_home(ctypes.byref(a), ops._stream()), 'rsa_shard_home')
        return out

    @ops._on_device
    def scatter_slots(self, dpos, dneg, slot_of, n_slots):
        """d loss/d score of a loss evaluated outside ``home`` -> routed order (dropped elements send nothing)."""
        B = dpos.numel()
        n = dneg.numel() // max(B, 1)
        d_send = torch.empty(n_slots, dtype=torch.float32, device=dpos.device)
        nat.check(nat.lib().rsa_shard_scatter_slots(ptr(ops._need(dpos, torch.float32, 'dpos')), ptr(ops._need(dneg, torch.float32, 'dneg')),
                                                    ptr(slot_of), B, n, ptr(d_send), ops._stream()), 'rsa_shard_scatter_slots')
        return d_send

    @ops._on_device
    def backward_segments(self, state, item_local, q_all, recv_keys, n_seg, stride, d_owner, item_grad_local, qgrad_all,
                          item_pad_row=-1, item_scale=None):
        """Owner side of the backward on received segments: item_grad_local[row] += s * d * q_all[qidx];
        qgrad_all[qidx] += g * d * item_local[row], with g = 0 when any rank dropped an element in this step (else 1)
        and s = g * item_scale (item_scale: device scalar, -lr for in-place SGD; default 1)."""
        m = n_seg * stride
        if m == 0:
            return
        scale = state['scale']
        if item_local.shape[1] in (64, 128, 256):
            # ONE call (rsa_shard_backward_segments): the slots radix-sorted by row and by query straight from the segments,
            # one walk over the query runs that reads every item row once (query-gradient partials in registers, rows a
            # single slot touches updated in place), the sorted apply pass for the rows several slots share
            a = nat.ShardBackwardArgs()
            a.item_local, a.n_rows, a.dim = ptr(item_local), item_local.shape[0], item_local.shape[1]
            a.q_all, a.n_query_rows = ptr(ops._need(q_all, torch.float32, 'q_all')), q_all.shape[0]
            a.keys, a.n_segments, a.stride = ptr(recv_keys), int(n_seg), int(stride)
            a.d_owner = ptr(ops._need(d_owner, torch.float32, 'd_owner'))
            a.item_target, a.item_scale = ptr(item_grad_local), ptr(item_scale)
            a.step_dropped, a.scale_out, a.qgrad_all = ptr(state['step_dropped']), ptr(scale), ptr(qgrad_all)
            a.item_pad_row = int(item_pad_row)
            nbytes = int(nat.lib().rsa_shard_backward_workspace_bytes(int(n_seg), int(stride), q_all.shape[0]))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=recv_keys.device)
            a.workspace, a.workspace_bytes = ptr(ws), nbytes
            nat.check(nat.lib().rsa_shard_backward_segments(ctypes.byref(a), ops._stream()), 'rsa_shard_backward_segments')
            return
        rows = torch.empty(m, dtype=torch.int64, device=recv_keys.device)
        qidx = torch.empty(m, dtype=torch.int64, device=recv_keys.device)
        nat.check(nat.lib().rsa_shard_unpack_segments(ptr(recv_keys), n_seg, stride, ptr(rows), ptr(qidx), ptr(item_scale),
                                                      ptr(state['step_dropped']), ptr(scale), ops._stream()),
                  'rsa_shard_unpack_segments')
        if item_scale is not None:
            raise NotImplementedError('in-place item update needs embed_dim in {64, 128, 256}')
        d_owner = torch.where(rows >= 0, d_owner * scale[1], torch.zeros((), device=d_owner.device))
        rows, qidx = rows.clamp_(min=0), qidx.clamp_(min=0)      # empty slots (d = 0 there): any valid row
        ops.fused_backward(item_local, q_all, rows.view(m, 1), d_owner.view(m, 1), query_index=qidx,
                           dense_item_grad=False, want_query_grad=False, item_grad_out=item_grad_local,
                           query_table_grad=qgrad_all, query_table_pad_row=-1, item_pad_row=item_pad_row)

    # -- the stock BPR step evaluated on the owners (rsa_shard_owner_bpr_forward / _finish) ---------------------------------
    OWNER_DIMS = (64, 128, 256)

    @ops._on_device
    def pos_scores(self, item_local, q_all, pos_rows):
        """[Q] scores of the positives this rank owns (pos_rows >= 0), 0 for the others."""
        out = torch.empty(pos_rows.numel(), dtype=torch.float32, device=q_all.device)
        nat.check(nat.lib().rsa_shard_pos_score(ptr(item_local), item_local.shape[0], item_local.shape[1], ptr(q_all), q_all.shape[0],
                                                ptr(ops._need(pos_rows, torch.int64, 'pos_rows')), ptr(out), None, 0, 1, 0,
                                                ops._stream()), 'rsa_shard_pos_score')
        return out

    @ops._on_device
    def pos_rows_and_scores(self, item_local, q_all, pos_all, plan, rank):
        """The same from the gathered GLOBAL ids, in one launch: (pos_rows [Q] -- local row of every positive this rank
        owns, -1 for the others --, scores [Q])."""
        Q = pos_all.numel()
        rows = torch.empty(Q, dtype=torch.int64, device=q_all.device)
        out = torch.empty(Q, dtype=torch.float32, device=q_all.device)
        nat.check(nat.lib().rsa_shard_pos_score(ptr(item_local), item_local.shape[0], item_local.shape[1], ptr(q_all), q_all.shape[0],
                                                ptr(rows), ptr(out), ptr(ops._need(pos_all, torch.int64, 'pos_all')), plan.rows_arg,
                                                plan.world, int(rank), ops._stream()), 'rsa_shard_pos_score')
        return rows, out

    @ops._on_device
    def owner_bpr_forward(self, state, item_local, q_all, recv_keys, n_seg, stride, pos_rows, pos_score, n, mean_den,
                          item_target, item_scale, qgrad_all, item_pad_row=-1, keys_grouped=False):
        """Scores, loss terms and gradients of the received negatives in ONE pass over their rows (see the header);
        -> ctx for ``owner_bpr_finish`` with ``dsum_part [Q]`` and ``loss_part`` (this rank's share of the mean loss)."""
        ctx = self._owner_ctx(state, item_local, q_all.shape[0], recv_keys, n_seg, stride, pos_rows, n, mean_den, item_target,
                              item_scale, item_pad_row, keys_grouped)
        return self._owner_launch(ctx, q_all, pos_score, qgrad_all, parts=0)

    def _owner_ctx(self, state, item_local, Q, recv_keys, n_seg, stride, pos_rows, n, mean_den, item_target, item_scale,
                   item_pad_row, keys_grouped):
        dev = item_local.device
        a = nat.ShardOwnerBprArgs()
        a.item_local, a.n_rows, a.dim, a.num_neg = ptr(item_local), item_local.shape[0], item_local.shape[1], int(n)
        a.n_query_rows = int(Q)
        a.keys, a.n_segments, a.stride = ptr(recv_keys), int(n_seg), int(stride)
        keep = {'pos_rows': ops._need(pos_rows, torch.int64, 'pos_rows'),
                'd_slots': torch.empty(n_seg * stride + Q, dtype=torch.float32, device=dev),
                'dsum_part': torch.empty(Q, dtype=torch.float32, device=dev),
                'loss_part': torch.empty((), dtype=torch.float32, device=dev),
                'tensors': [item_local, recv_keys, item_target, item_scale]}
        a.pos_rows, a.mean_den = ptr(keep['pos_rows']), int(mean_den)
        a.item_target, a.item_scale = ptr(item_target), ptr(item_scale)
        a.step_dropped, a.overflow_sticky, a.scale_out = ptr(state['step_dropped']), ptr(state['overflow']), ptr(state['scale'])
        a.d_slots, a.dsum_part, a.loss_part = ptr(keep['d_slots']), ptr(keep['dsum_part']), ptr(keep['loss_part'])
        a.reduce_scratch, a.item_pad_row, a.keys_grouped = ptr(ops._scratch()), int(item_pad_row), int(bool(keys_grouped))
        nbytes = int(nat.lib().rsa_shard_backward_workspace_bytes(int(n_seg), int(stride), int(Q)))
        keep['ws'] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        a.workspace, a.workspace_bytes = ptr(keep['ws']), nbytes
        keep['args'] = a
        return keep

    def _owner_launch(self, ctx, q_all, pos_score, qgrad_all, parts):
        a = ctx['args']
        if parts != 1:
            ctx['pos_score'] = ops._need(pos_score, torch.float32, 'pos_score')
            ctx['tensors'] += [q_all, qgrad_all]
            a.q_all, a.pos_score, a.qgrad_all = ptr(ops._need(q_all, torch.float32, 'q_all')), ptr(ctx['pos_score']), ptr(qgrad_all)
        a.forward_parts = int(parts)
        nat.check(nat.lib().rsa_shard_owner_bpr_forward(ctypes.byref(a), ops._stream()), 'rsa_shard_owner_bpr_forward')
        return ctx

    @ops._on_device
    def owner_bpr_prepare(self, state, item_local, Q, recv_keys, n_seg, stride, pos_rows, n, mean_den, item_target, item_scale,
                          item_pad_row=-1, keys_grouped=False):
        """The part of ``owner_bpr_forward`` that reads only the received keys and the positives' rows -- sort by row, solo
        classification, the queries' runs (``forward_parts = 1``) -- on the current stream; -> ctx for ``owner_bpr_walk``."""
        ctx = self._owner_ctx(state, item_local, Q, recv_keys, n_seg, stride, pos_rows, n, mean_den, item_target, item_scale,
                              item_pad_row, keys_grouped)
        return self._owner_launch(ctx, None, None, None, parts=1)

    @ops._on_device
    def owner_bpr_walk(self, ctx, q_all, pos_score, qgrad_all):
        """The rest of the forward over a prepared ctx (``forward_parts = 2``): update scales, the pass over the rows."""
        return self._owner_launch(ctx, q_all, pos_score, qgrad_all, parts=2)

    @ops._on_device
    def owner_bpr_finish(self, ctx, dsum_all, parts=0):
        """``parts``: 0 = the whole finish; 1 = the positives only (``qgrad_all`` is complete afterwards); 2 = the shared rows'
        sorted apply pass (after a call with 1)."""
        ctx['args'].finish_parts = int(parts)
        nat.check(nat.lib().rsa_shard_owner_bpr_finish(ctypes.byref(ctx['args']), ptr(ops._need(dsum_all, torch.float32, 'dsum_all')),
                                                       ops._stream()), 'rsa_shard_owner_bpr_finish')

    @ops._on_device
    def owner_ssm_forward(self, state, item_local, q_all, recv_keys, n_seg, stride, pos_rows, n, mean_den, item_target, item_scale,
                          logq_rows=None, item_pad_row=-1, keys_grouped=False):
        """Phase 1 of SampledSoftmax on the owners (rsa_shard_owner_ssm_forward): one walk over the received negatives' rows
        -> ctx with ``run_max [Q]``, ``run_sum [Q]`` (and, kept for phase 2, ``run_acc [Q, d]`` and z per slot)."""
        Q, dev = q_all.shape[0], item_local.device
        ctx = self._owner_ctx(state, item_local, Q, recv_keys, n_seg, stride, pos_rows, n, mean_den, item_target, item_scale,
                              item_pad_row, keys_grouped)
        a = ctx['args']
        ctx['run_max'] = torch.empty(Q, dtype=torch.float32, device=dev)
        ctx['run_sum'] = torch.empty(Q, dtype=torch.float32, device=dev)
        ctx['run_acc'] = torch.empty(Q, item_local.shape[1], dtype=torch.float32, device=dev)
        ctx['tensors'] += [q_all, logq_rows]
        a.q_all = ptr(ops._need(q_all, torch.float32, 'q_all'))
        a.logq_rows = ptr(ops._need_opt(logq_rows, torch.float32, 'logq_rows'))
        a.run_max, a.run_sum, a.run_acc = ptr(ctx['run_max']), ptr(ctx['run_sum']), ptr(ctx['run_acc'])
        nat.check(nat.lib().rsa_shard_owner_ssm_forward(ctypes.byref(a), ops._stream()), 'rsa_shard_owner_ssm_forward')
        return ctx

    @ops._on_device
    def owner_ssm_finish(self, ctx, lse_all, z_pos_all, qgrad_all):
        """Phase 2 (rsa_shard_owner_ssm_finish): d per slot, the query-gradient partials from the phase-1 accumulators, the
        positives' terms, the sorted apply pass over every touched row."""
        a = ctx['args']
        ctx['pos_score'] = ops._need(z_pos_all, torch.float32, 'z_pos_all')
        ctx['tensors'] += [qgrad_all, lse_all]
        a.pos_score, a.qgrad_all = ptr(ctx['pos_score']), ptr(ops._need(qgrad_all, torch.float32, 'qgrad_all'))
        nat.check(nat.lib().rsa_shard_owner_ssm_finish(ctypes.byref(a), ptr(ops._need(lse_all, torch.float32, 'lse_all')),
                                                       ops._stream()), 'rsa_shard_owner_ssm_finish')

    # -- exact (variable-split) exchange ------------------------------------------------------------------------------
    def gather_rows(self, table, ids):
        if ids.numel() == 0:
            return table.new_empty(*ids.shape, table.shape[1])
        return ops.embedding_gather(table, ids)

    def count(self, pos, neg, plan):
        counts = torch.empty(plan.world, dtype=torch.int32, device=pos.device)
        n = neg.shape[1]
        nat.check(nat.lib().rsa_shard_count(ptr(pos), ptr(neg), pos.numel(), n, plan.rows_arg, plan.world,
                                            ptr(counts), ops._stream()), 'rsa_shard_count')
        return counts

    def route(self, pos, neg, plan, query_base, starts):
        B, n = neg.shape
        numel = B * (n + 1)
        keys = torch.empty(numel, dtype=torch.int64, device=pos.device)
        positions = torch.empty(numel, dtype=torch.int64, device=pos.device)
        cursor = starts.to(device=pos.device, dtype=torch.int32).clone()
        nat.check(nat.lib().rsa_shard_route(ptr(pos), ptr(neg), B, n, plan.rows_arg, plan.world, int(query_base),
                                            ptr(cursor), ptr(keys), ptr(positions), ops._stream()), 'rsa_shard_route')
        return keys, positions

    def score_keys(self, item_local, q_all, keys):
        return ops.score_packed_keys(item_local, q_all, keys)

    def scatter(self, scores, positions, numel):
        dst = torch.zeros(numel, dtype=torch.float32, device=scores.device)
        nat.check(nat.lib().rsa_scatter_f32(ptr(scores), ptr(positions), scores.numel(), ptr(dst), ops._stream()),
                  'rsa_scatter_f32')
        return dst

    def gather(self, src, positions):
        dst = torch.empty(positions.numel(), dtype=torch.float32, device=src.device)
        nat.check(nat.lib().rsa_gather_f32(ptr(src), ptr(positions), positions.numel(), ptr(dst), ops._stream()),
                  'rsa_gather_f32')
        return dst

    def backward_keys(self, item_local, q_all, keys, dscore, item_grad_local, qgrad_all, item_pad_row=-1, item_scale=None):
        """Owner side of the backward (exact exchange): item_grad_local[row] += d * q_all[qidx]; qgrad_all[qidx] += d *
        item_local[row].  ``item_scale`` (device scalar): multiply the item-side update by it -- with
        ``item_grad_local is item_local`` and item_scale = -lr this is plain SGD applied in place (the query-side pass,
        which reads the rows, runs first)."""
        m = keys.numel()
        if m == 0:
            return
        rows = torch.empty(m, dtype=torch.int64, device=keys.device)
        qidx = torch.empty(m, dtype=torch.int64, device=keys.device)
        nat.check(nat.lib().rsa_shard_unpack(ptr(keys), m, ptr(rows), ptr(qidx), ops._stream()), 'rsa_shard_unpack')
        if item_local.shape[1] in (64, 128, 256):
            ops.scatter_rows_sorted(qgrad_all, item_local, qidx.view(m, 1), dscore.view(m, 1), query_index=rows, pad_row=-1)
            ops.scatter_rows_sorted(item_grad_local, q_all, rows.view(m, 1), dscore.view(m, 1), query_index=qidx,
                                    upstream=item_scale, pad_row=item_pad_row)
            return
        if item_scale is not None:
            raise NotImplementedError('in-place item update needs embed_dim in {64, 128, 256}')
        rows, qidx = rows.clamp_(min=0), qidx.clamp_(min=0)
        ops.fused_backward(item_local, q_all, rows.view(m, 1), dscore.view(m, 1), query_index=qidx,
                           dense_item_grad=False, want_query_grad=False, item_grad_out=item_grad_local,
                           query_table_grad=qgrad_all, query_table_pad_row=-1, item_pad_row=item_pad_row)

    def apply_rows(self, table, ids, rows, scale, pad_row=0, gate=None):
        """table[ids[e]] += gate * scale * rows[e] for ids != pad_row (pad_row < 0: every row), duplicates summed in sorted
        order without atomics for embed_dim in {64, 128, 256}: every replica that applies the same (ids, rows) ends up
        with the same bits.  Other dims: ``index_add_`` (float atomics: equal up to summation order).  ``gate``: an optional
        device scalar (the step's overflow gate, 0 or 1)."""
        m = ids.numel()
        if m == 0:
            return
        d = rows.shape[1]
        if d in (64, 128, 256):
            ops.scatter_rows_sorted(table, rows, ids.view(m, 1), torch.full((m, 1), float(scale), device=rows.device),
                                    query_index=torch.arange(m, device=ids.device), pad_row=pad_row, upstream=gate)
        else:
            keep = (ids != pad_row) & (ids >= 0)
            upd = rows[keep] * float(scale)
            table.index_add_(0, ids[keep], upd if gate is None else upd * gate)

    def full_partial(self, item_local, q_all, k, want_lse, has_pad_row):
        """This shard's part of the full-catalog pass (BASELINE.json configs[4] sharded):
        logsumexp over the local rows and the local top-k as (values, 1-based LOCAL row numbers, i.e.
        row r of ``item_local`` -> r + 1 without a padding row, r with one)."""
        _, lse, tv, ti = ops.fullscore(item_local, q_all, want_lse=want_lse, k=k, items_without_pad=not has_pad_row)
        return lse, tv, ti

    def merge_lse(self, parts):
        """parts [B, G] per-shard logsumexp -> [B]."""
        return ops.row_lse(parts.contiguous())[0]

    def merge_topk(self, vals, ids, k):
        """vals/ids [B, G*k] -> exact global top-k, ties -> smaller COLUMN (the caller hands the candidates over in an order
        in which that is the smaller id: shard-major for contiguous row blocks, id-sorted for interleaved rows)."""
        if k > ops.FULLSCORE_MAX_K:          # the wide correctness path (k + |history| beyond the in-kernel select)
            v, cols = torch.sort(vals, dim=1, descending=True, stable=True)
            v, cols = v[:, :k].contiguous(), cols[:, :k]
        else:
            v, cols = ops.row_topk(vals.contiguous(), k)
        return v, torch.gather(ids, 1, cols)


_GATHER_GROUPS = {}


def _gather_group(dist):
    """One extra communicator over all ranks per process (and per torch.distributed look-alike), shared by every
    ShardedItemTable: a communicator costs device buffers and a collective set-up, tables are cheap.  Only a WEAK
    reference is kept: torch.distributed owns the group until destroy_process_group(); a strong reference here kept
    the (gloo) backend's threads alive past that call and the process aborted at interpreter shutdown now and then."""
    world = getattr(getattr(dist, 'group', None), 'WORLD', None)     # a re-initialised process group is a new object
    key = (id(dist), id(world))
    hit = _GATHER_GROUPS.get(key)
    if hit is not None:
        if hit == 'none':                      # a look-alike whose new_group() returns None
            return None
        group = hit()
        if group is not None:
            return group
    _GATHER_GROUPS.clear()                     # groups of an earlier initialisation are dead
    group = dist.new_group()
    _GATHER_GROUPS[key] = 'none' if group is None else weakref.ref(group)
    return group


class ShardOverflow(RuntimeError):
    """Routed elements did not fit their owner segment (``ShardedItemTable.check_overflow``): the affected steps changed no
    weight; the capacity is recalibrated on the next step.  Its own type so that a deferred poll can downgrade THIS to a
    message and nothing else (a HIP / copy failure raised while polling must propagate)."""


class ShardedItemTable:
    def __init__(self, item_local, plan, rank, dist, backend=None, group=None, exchange='fixed', slack=1.08,
                 margin=4096, check_every=16, sample_seed=2022, chunks=1, force_collectives=False, owner_loss=True,
                 deterministic=True, rows_share=None):
        """``chunks`` > 1 (fixed-capacity exchange only): the step's queries are cut into that many contiguous
        slices, routed by ONE launch, whose exchanges are issued asynchronously, so that slice c+1's key all-to-all and
        slice c-1's score all-to-all travel over xGMI while slice c is being scored (see ``_fixed_step``)."""
        self.item_local, self.plan, self.rank, self.dist = item_local, plan, int(rank), dist
        self.chunks = max(1, int(chunks))
        self.owner_loss = bool(owner_loss)      # stock BPR training steps are evaluated on the owners (bpr_step_on_owners)
        # deterministic (the default since round 6: it is free -- 1164.5 vs 1168.7 us tracked at configs[3]'s per-GPU shape): the
        # router takes three launches instead of one (count, prefix over the workgroups, route) and no atomic decides where a
        # key lands -- a step's segments, and every sum the owners form in slot order, are then bit-identical run to run and
        # "G ranks == 1 rank" holds with torch.equal (False: slots follow returning atomics, steps agree to fp32 rounding only)
        self.deterministic = bool(deterministic)
        self.rows_share = None if rows_share is None else float(rows_share)      # see _lookup_rows_fixed
        self.group_by_query = True              # ... with query-grouped routing where the shape allows (no sort by query there)
        self._solo = plan.world == 1 and not force_collectives
        self.backend = backend if backend is not None else HipBackend()
        self.group = group
        if exchange not in ('fixed', 'exact'):
            raise ValueError("exchange must be 'fixed' or 'exact'")
        self.exchange, self.slack, self.margin, self.check_every = exchange, float(slack), int(margin), int(check_every)
        self._cap, self._steps, self._poll, self._poll_due = {}, 0, None, 0
        self.state = self.backend.new_state(item_local.device)
        # ONE sampler stream for the whole job: same seed on every rank, advanced in lock-step (see module docstring)
        self.sample_generator = self.backend.make_generator(sample_seed, item_local.device)
        # the query all-gather runs on its OWN communicator: collectives of one communicator execute in issue
        # order on one stream, so on the main group the key / score exchanges would queue behind the (at n = 64)
        # much larger gather instead of overlapping with it
        self.gather_group = _gather_group(dist) if (group is None and plan.world > 1) else group
        if item_local.shape[0] != plan.n_local(rank):
            raise ValueError(f'rank {rank} must hold its {plan.n_local(rank)} rows of the item table ({plan.layout} layout), '
                             f'got {item_local.shape[0]}')

    # -- collectives (RCCL through torch.distributed) -------------------------------------------
    # With ONE rank every collective of the step is the identity (the all-gather of one block, an all-to-all with
    # itself, a reduce-scatter of one part): unless ``force_collectives`` asks for the calls anyway (tests, and the
    # bench's protocol-cost figure) they are skipped -- no copy through the communicator, no launch.
    def _all_reduce_max(self, t):
        if not self._solo:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)

    def _all_gather_rows(self, x):
        return self._all_gather_rows_start(x)()

    def _all_gather_rows_start(self, x):
        """Start the all-gather of the query block on the communicator's own stream and return a function
        that waits for it.  At B = 65536 queries/GPU the block is 33.5 MB per rank -- the largest message of
        the step (7 x 33.5 MB arrive per GPU over xGMI, vs 8 + 4 bytes per triplet for keys and scores) -- and
        nothing before the owner-side scoring needs it, so it flies under sampling and routing."""
        x = x.contiguous()
        if self._solo:
            return lambda: x
        out = torch.empty(self.plan.world * x.shape[0], *x.shape[1:], dtype=x.dtype, device=x.device)
        work = self.dist.all_gather_into_tensor(out, x, group=self.gather_group, async_op=True)

        def wait():
            work.wait()
            return out
        wait.keep = x           # the source must stay alive until the collective has run
        return wait

    def _exchange_counts(self, counts):
        send = counts.to(torch.int64)
        recv = torch.empty_like(send)
        if self._solo:
            recv = send
        else:
            self.dist.all_to_all_single(recv, send, group=self.group)
        return [int(v) for v in send.tolist()], [int(v) for v in recv.tolist()]

    def _all_to_all(self, x, recv_counts=None, send_counts=None, out=None):
        """Variable split (host lists) or, with no counts, the equal split of the fixed-capacity exchange."""
        if self._solo:
            return x if out is None else out.copy_(x)
        if recv_counts is None:
            out = torch.empty_like(x) if out is None else out
            self.dist.all_to_all_single(out, x, group=self.group)
            return out
        if out is None:
            out = torch.empty(sum(recv_counts), *x.shape[1:], dtype=x.dtype, device=x.device)
        self.dist.all_to_all_single(out, x, output_split_sizes=recv_counts, input_split_sizes=send_counts,
                                    group=self.group)
        return out

    def _all_to_all_start(self, x, out=None):
        """Equal-split all-to-all issued asynchronously (it runs on the communicator's stream once everything queued
        on the current stream so far has finished); returns wait() -> the received tensor, ordered after the transfer
        on the current stream."""
        if self._solo:
            res = x if out is None else out.copy_(x)
            return lambda: res
        out = torch.empty_like(x) if out is None else out
        work = self.dist.all_to_all_single(out, x, group=self.group, async_op=True)

        def wait():
            if work is not None:
                work.wait()
            return out
        wait.keep = x
        return wait

    def _reduce_scatter_rows(self, x, rows_per_rank):
        if self._solo:
            return x
        out = torch.empty(rows_per_rank, *x.shape[1:], dtype=x.dtype, device=x.device)
        self.dist.reduce_scatter_tensor(out, x.contiguous(), group=self.group)
        return out

    # -- fixed-capacity bookkeeping -------------------------------------------------------------------
    def _capacity(self, key, largest, store=None):
        """Capacity of one (slice, owner) segment from the exact counts of a step: the largest count of ANY rank (so
        that every rank uses the same equal split) plus slack.  One small all-reduce + read-back, on the calibration
        step only."""
        B, n, C = key
        m = torch.tensor([int(largest)], dtype=torch.int64, device=self.item_local.device)
        self._all_reduce_max(m)
        S = int(getattr(self.backend, 'BANKS', 1))
        cap = int(int(m.item()) * self.slack) + (self.margin + S - 1) // S
        cap = min(B * (n + 1), (cap + 255) // 256 * 256)
        store = key if store is None else store
        self._cap[store] = max(cap, 1)
        return self._cap[store]

    def check_overflow(self, block=True):
        """Raise if any routed element of ANY rank found its owner's segment full since the last check.  The sticky
        count is job-wide on every rank (the segment headers carry every source's dropped count to every owner), so
        all ranks raise at the same step without a collective.  The steps in which it happened changed no weight
        (their update scale was 0 on the device)."""
        self._poll = self._poll or self.backend.flag_read_async(self.state['overflow'])
        v = self._poll()
        while v is None:
            v = self._poll()
        self._poll = None
        if v:
            self._cap.clear()               # recalibrate on the next step
            self.state['overflow'].zero_()
            raise ShardOverflow(f'sharded exchange: {v} elements (all ranks) did not fit their owner segment (capacity slack '
                               f'{self.slack}); the affected steps were skipped (their weight updates were scaled by 0) '
                               "-- raise `slack` / `margin` or use exchange='exact' for id distributions that drift this fast")

    def take_overflow(self):
        """The message of an overflow the deferred poll found since the last call (None: none) -- see ``defer_overflow``."""
        msg, self.overflow_message = getattr(self, 'overflow_message', None), None
        return msg

    LAG = 4          # steps between starting the read-back of the overflow word and looking at it

    def _after_fixed_step(self):
        """Every ``check_every`` steps the overflow count starts its way to the host; LAG steps later -- long after
        the copy has landed, so without a stall -- every rank looks at it, at the SAME step."""
        self._steps += 1
        if self.check_every <= 0:
            return
        if self._poll is not None and self._steps >= self._poll_due:
            if getattr(self, 'defer_overflow', False):
                # a training loop asked to be TOLD instead of interrupted: the poll lags LAG steps behind the overflowed
                # step, so raising here would cut a healthy step in half -- its item rows already updated in place, its
                # query-row exchange, tower backward, all-reduce and optimizer step not run.  The message waits
                # in `overflow_message` for `take_overflow()` at the END of the step.
                try:
                    self.check_overflow()
                except ShardOverflow as err:          # (only this: any other error raised while polling propagates)
                    self.overflow_message = str(err)
            else:
                self.check_overflow()
        if self._poll is None and self._steps % self.check_every == 0:
            self._poll = self.backend.flag_read_async(self.state['overflow'])
            self._poll_due = self._steps + self.LAG

    # -- the step ---------------------------------------------------------------------------------
    def _route_and_exchange(self, pos, n, spec, neg, fused_loss, want_ids, want_logp):
        """The part of the fixed step that reads no weight and no query: calibration (first call), the routing launch, the key
        exchange of every slice.  -> (route, slices, [wait functions of the key exchanges])."""
        be, st, plan = self.backend, self.state, self.plan
        B = pos.numel()
        C = self.chunks if (self.chunks > 1 and B % self.chunks == 0) else 1
        S = int(getattr(be, 'BANKS', 1))           # segments per (slice, owner): every consumer just sees G * S segments
        # (the capacity is calibrated per id SOURCE: ids a plugin hands in -- popularity-skewed, say -- must not reuse the
        # capacity of the in-kernel uniform draw)
        key = (B, n, C) if spec is not None else (B, n, C, 'given')
        cap = self._cap.get(key)
        if cap is None:
            # calibration: exact per-segment counts of the very draw this step routes (the generator is not advanced),
            # largest over slices, owners, banks and ranks
            counts = be.sample_route(st, plan, self.rank, pos, n, C, 0, spec, self.sample_generator, neg=neg, count_only=True,
                                     banks=S)
            cap = self._capacity(key[:3], int(counts.max()), store=key)
        # the sampler's log-probabilities: BPRLoss ignores them (loss_func.py:55-59), everything else gets them
        r = be.sample_route(st, plan, self.rank, pos, n, C, cap, spec, self.sample_generator, neg=neg,
                            want_ids=want_ids, want_logp=want_logp and fused_loss != 'bpr', banks=S, deterministic=self.deterministic)
        per = plan.world * S * r['stride']
        send = r['send']
        if C == 1:
            return r, C, [lambda: self._all_to_all(send)]
        return r, C, [self._all_to_all_start(send[c * per:(c + 1) * per]) for c in range(C)]

    def prepare_forward(self, pos, n, sampler, fused_loss=None, want_ids=True, want_logp=True):
        """One batch ahead (fixed exchange, in-kernel sampler): the negatives of ``pos``'s batch drawn and routed and the key
        exchange done, on a SECOND stream when the ids live on a GPU -- under whatever the main stream is doing (the step in
        front).  -> ticket for ``forward_queries(..., ticket=)`` / ``sample_and_score(..., ticket=)`` with the same ``n`` /
        ``fused_loss``.  Tickets are consumed once each, in the order they were prepared (the draws follow the job's sample
        stream in that order); nothing else may route through this table in between."""
        spec = self.backend.sampler_spec(sampler) if self.exchange == 'fixed' else None
        if spec is None:
            raise NotImplementedError("prepare_forward: exchange='fixed' with one of this package's in-kernel samplers")

        def issue():
            r, C, waits = self._route_and_exchange(pos, n, spec, None, fused_loss, want_ids, want_logp)
            return self._stamp({'B': pos.numel(), 'n': int(n), 'C': C, 'fused_loss': fused_loss, 'route': r,
                                'recv': [w() for w in waits]}, pos, want_ids=bool(want_ids), want_logp=bool(want_logp))
        return self._on_second_stream(issue, pos)

    # -- ticket identity: a ticket is consumed once, in preparation order, by the batch it was prepared for -------------------
    def _stamp(self, ticket, pos, **what):
        self._tickets_issued = getattr(self, '_tickets_issued', 0) + 1
        ticket.update(seq=self._tickets_issued, pos_key=(pos.data_ptr(), tuple(pos.shape)), **what)
        return ticket

    def _claim(self, ticket, pos, who, **expect):
        """Refuse a ticket stepped out of order, a second time, with another batch's positives or with other output
        requests than it was prepared with -- each of which would silently train / score the new queries against the
        prepared batch's positives and negatives."""
        done = getattr(self, '_tickets_claimed', 0)
        if not isinstance(ticket.get('seq'), int) or ticket['seq'] <= done:       # (an abandoned ticket may be skipped over)
            raise ValueError(f"{who}: tickets are consumed once each, in the order they were prepared (this is ticket "
                             f"#{ticket.get('seq')}; #{done} has been consumed already)")
        nxt = ticket['seq']
        if pos is not None and ticket.get('pos_key') != (pos.data_ptr(), tuple(pos.shape)):
            raise ValueError(f'{who}: the ticket was prepared for another batch (its positives are not the tensor passed here)')
        for k, v in expect.items():
            if ticket.get(k) != v:
                raise ValueError(f'{who}: the ticket was prepared with {k}={ticket.get(k)!r}, this call asks for {k}={v!r}')
        self._tickets_claimed = nxt

    MAX_TICKETS_AHEAD = 3      # preparations the host may have in flight (see _on_second_stream)

    def _on_second_stream(self, issue, like):
        """``issue()`` on this table's second stream (CPU tensors: in place); the ticket it returns gets a ``ready`` event and
        its tensors are handed to the main stream."""
        if not like.is_cuda:
            with torch.no_grad():
                return dict(issue(), ready=None)
        dev = like.device
        if getattr(self, '_second', None) is None:
            self._second = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        # The HOST must not run further ahead of the GPU than a few tickets: a ticket's buffers are allocated on the second
        # stream and freed after the main stream has consumed them, so the caching allocator can hand a block out again only
        # once the consuming step has COMPLETED -- a host that enqueues hundreds of steps ahead (the Python of a step costs a
        # third of its GPU time) gets a fresh device allocation for every ticket (measured: 15 -> 73 GB reserved in 1.5 s of
        # look-ahead steps at the configs[3] shape; on a full HBM, allocator retries that synchronise the device).  Waiting
        # for the preparation issued MAX_TICKETS_AHEAD calls ago costs nothing on the GPU (it has that many steps queued).
        pend = self.__dict__.setdefault('_ready_pending', collections.deque())
        while len(pend) >= self.MAX_TICKETS_AHEAD:
            pend.popleft().synchronize()
        self._second.wait_stream(main)                # the batch tensors may have been produced on the main stream
        with torch.no_grad(), torch.cuda.stream(self._second):
            ticket = issue()
            ready = torch.cuda.Event()
            ready.record(self._second)
        _record_stream_all(ticket, main)              # allocated on the second stream, consumed (and freed) on the main one
        ticket['ready'] = ready
        pend.append(ready)
        return ticket

    def _fixed_step(self, q_gather, pos, n, spec, neg=None, keep_route=False, fused_loss=None, mean_den=None,
                    log_pos=None, log_neg=None, want_ids=True, want_logp=True, want_scores=True, want_grad=False, ticket=None):
        """Version 2 of the fixed-capacity step.  Issue order on the current stream: ONE routing launch for all
        ``chunks`` query slices (each slice's key all-to-all follows on the communicator's stream), then per slice:
        wait for its keys, score, start the score all-to-all; finally wait for all scores and run the home kernel
        once.  Collectives of one communicator run in issue order, so slice c's keys arrive while slice c-1 is being
        scored and its scores go back while slice c+1 is being scored.  Every rank issues the same sequence."""
        be, st, plan = self.backend, self.state, self.plan
        B, G = pos.numel(), plan.world
        S = int(getattr(be, 'BANKS', 1))
        if ticket is None:
            r, C, waits = self._route_and_exchange(pos, n, spec, neg, fused_loss, want_ids, want_logp)
        else:
            if (ticket['B'], ticket['n'], ticket['fused_loss']) != (B, int(n), fused_loss):
                raise ValueError('the ticket was prepared for another batch shape / loss')
            self._claim(ticket, pos, 'forward_queries(ticket=)', want_ids=bool(want_ids), want_logp=bool(want_logp))
            if ticket.get('ready') is not None:
                torch.cuda.current_stream(pos.device).wait_event(ticket['ready'])
            r, C = ticket['route'], ticket['C']
            waits = [(lambda rk=rk: rk) for rk in ticket['recv']]
        GS = G * S
        stride, per = r['stride'], GS * r['stride']
        scores_home = torch.empty(C * per, dtype=torch.float32, device=r['send'].device)
        recv_keys = []
        if C == 1:
            q_all = q_gather()
            rk = waits[0]()
            recv_keys.append(rk)
            if self._solo:
                be.score_segments(st, self.item_local, q_all, rk, GS, stride, first=True, out=scores_home)
            else:
                self._all_to_all(be.score_segments(st, self.item_local, q_all, rk, GS, stride, first=True), out=scores_home)
        else:
            q_all = q_gather()
            back = []
            for c, w in enumerate(waits):
                rk = w()
                recv_keys.append(rk)
                home_c = scores_home[c * per:(c + 1) * per]
                if self._solo:
                    be.score_segments(st, self.item_local, q_all, rk, GS, stride, first=c == 0, out=home_c)
                else:
                    sc = be.score_segments(st, self.item_local, q_all, rk, GS, stride, first=c == 0)
                    back.append(self._all_to_all_start(sc, out=home_c))
            for w in back:
                w()
        if log_pos is None:
            log_pos, log_neg = r.get('log_pos_prob'), r.get('log_neg_prob')
        out = be.home(scores_home, r['slot_of'], B, n, loss=fused_loss, pos_logp=log_pos if fused_loss == 'ssm' else None,
                      neg_logp=log_neg if fused_loss == 'ssm' else None, mean_den=mean_den, want_scores=want_scores,
                      want_grad=want_grad, want_dsend=keep_route and fused_loss is not None)
        self._after_fixed_step()
        out['neg_ids'], out['log_pos_prob'], out['log_neg_prob'] = r.get('neg_ids'), log_pos, log_neg
        if keep_route:
            # the step's job-wide dropped count travels with the route: a forward issued between this step's forward and
            # its backward (an evaluation, a second table user) overwrites the table's word
            out['route'] = {'B': B, 'n': n, 'C': C, 'GS': GS, 'stride': stride, 'q_all': q_all, 'slot_of': r['slot_of'],
                            'recv_keys': recv_keys, 'd_send': out.pop('d_send', None), 'dropped': st['step_dropped'].clone()}
        return out

    def _all_reduce_sum(self, t):
        if not self._solo:
            self.dist.all_reduce(t, group=self.group)
        return t

    def _all_reduce_max(self, t):
        if not self._solo:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return t

    def _logq_rows(self, sampler):
        """log-probability under ``sampler`` of the item of every LOCAL row ([rows_local] fp32; None for samplers whose
        log-probabilities are zero): the owner-side SampledSoftmax reads it per received negative."""
        prob = getattr(sampler, 'pop_prob', None)
        if prob is None:
            return None
        hit = getattr(self, '_logq_cache', None)
        if hit is not None and hit[0] is prob and hit[1] == prob._version:
            return hit[2]
        dev = self.item_local.device
        gid = self.plan.global_ids(self.rank, torch.arange(self.item_local.shape[0], device=dev))
        logq = torch.log(prob.to(dev)[gid.clamp(max=prob.numel() - 1)]).to(torch.float32).contiguous()
        self._logq_cache = (prob, prob._version, logq)
        return logq

    def ssm_owner_ok(self):
        """Can the stock SampledSoftmax step run in its owner-side form (``ssm_step_on_owners``)?"""
        return self.owner_loss_ok() and hasattr(self.backend, 'owner_ssm_forward')

    def ssm_step_on_owners(self, q, pos, n, sampler, item_grad_local, item_scale=None, want_ids=False):
        """The stock SampledSoftmax training step (loss_func.py:80-90, one positive per row) with the loss evaluated ON THE
        OWNERS of the negatives (include/recstudio_amd.h, rsa_shard_owner_ssm_forward / _finish).  The logsumexp of a query
        spans every owner of its negatives, so the owner pass has two phases around an 8-byte-per-query all-reduce:

          1. as ``bpr_step_on_owners``: queries and positive ids all-gathered, negatives drawn and routed (positives are not),
             the positives scored by their owners (4-byte-per-query all-reduce);
          2. phase 1 on every owner -- ONE walk over the rows of the negatives it received: z = score - log q per slot and,
             per query, (max z, sum exp(z - max), sum exp(z - max) * row);
          3. all-reduce of the maxima, then of the rescaled sums: every rank holds every query's logsumexp;
          4. phase 2: d loss/d score per slot, the query-gradient partials straight from the phase-1 accumulators (no second
             pass over the rows for them), the positives' terms, the sorted apply pass over every touched item row;
             reduce-scatter of the query gradients.

        No scores travel home and no score gradients travel back (8 instead of 16 bytes per triplet over xGMI), and the home
        kernel is gone.  -> (this rank's share of the global mean loss, d loss/d q [B, d], negative ids or None)."""
        be, st, plan = self.backend, self.state, self.plan
        B, G = pos.numel(), plan.world
        q_gather = self._all_gather_rows_start(q)
        pos_all, r, recv, neg_out = self._bpr_route(pos, n, sampler, want_ids)
        GS, stride = G * int(getattr(be, 'BANKS', 1)), r['stride']
        q_all = q_gather()
        if hasattr(be, 'pos_rows_and_scores'):
            pos_rows, pos_part = be.pos_rows_and_scores(self.item_local, q_all, pos_all, plan, self.rank)
        else:
            pos_rows = self._own_pos_rows(pos_all)
            pos_part = be.pos_scores(self.item_local, q_all, pos_rows)
        z_pos = self._all_reduce_sum(pos_part)
        prob = getattr(sampler, 'pop_prob', None)                         # (samplers without it: log-probabilities 0)
        if prob is not None:
            z_pos = z_pos - torch.log(prob.to(z_pos.device)[pos_all]).to(z_pos.dtype)     # loss_func.py:81, compute_item_p
        ctx = be.owner_ssm_forward(st, self.item_local, q_all, recv, GS, stride, pos_rows, n, B * G, item_grad_local, item_scale,
                                   logq_rows=self._logq_rows(sampler), item_pad_row=0 if self.rank == 0 else -1,
                                   keys_grouped=bool(r.get('grouped', False)))
        run_max = ctx['run_max']
        m = torch.maximum(self._all_reduce_max(run_max.clone()), z_pos)
        s = self._all_reduce_sum(ctx['run_sum'] * torch.exp(run_max - m))   # (run_max = -inf, run_sum = 0: no slot on this owner)
        lse = m + torch.log(s + torch.exp(z_pos - m))
        qgrad_all = torch.zeros_like(q_all)
        be.owner_ssm_finish(ctx, lse, z_pos, qgrad_all)
        dq = self._reduce_scatter_rows(qgrad_all, B)
        self._after_fixed_step()
        loss = (lse - z_pos)[self.rank * B:(self.rank + 1) * B].sum() / float(B * G)
        return loss, dq, neg_out

    def owner_loss_ok(self):
        """Can the stock BPR step run in its owner-side form (``bpr_step_on_owners``)?"""
        return (self.owner_loss and self.exchange == 'fixed' and self.chunks == 1 and hasattr(self.backend, 'owner_bpr_forward')
                and self.item_local.shape[1] in getattr(self.backend, 'OWNER_DIMS', (64, 128, 256)))

    def bpr_step_on_owners(self, q, pos, n, sampler, item_grad_local, item_scale=None, want_ids=False, defer_rows=False,
                           ticket=None):
        """The stock BPR training step with the loss evaluated ON THE OWNERS of the negatives (SURVEY.md 8e, steps 3-5 in one
        pass; include/recstudio_amd.h, rsa_shard_owner_bpr_forward): BPR's d loss/d neg needs only the query's positive
        score, so

          1. all-gather of the queries and of the positive ids; the negatives are drawn and routed as usual, the positives
             are NOT routed;
          2. the rank that owns a positive's row scores it; a 4-byte-per-query all-reduce gives every rank every positive's
             score;
          3. every owner evaluates score, loss term and gradient of the negatives it received while their rows are in
             registers -- query-gradient partials, in-place updates of the rows one element touches -- and the queries' sums
             of d; a second 4-byte-per-query all-reduce turns those into d loss/d pos;
          4. the positives' owners finish (their query-gradient term, their rows), the rows several elements touch go
             through the sorted apply pass; reduce-scatter of the query gradients.

        No scores travel home, no score gradients travel back (8 instead of 16 bytes per triplet over xGMI), and the item
        rows of a step are read once instead of twice.  -> (this rank's share of the global mean loss, d loss/d q [B, d],
        negative ids or None).  The shares of all ranks add up to the loss (each rank holds the terms of the negatives it
        owns).

        ``defer_rows``: the last part of step 4 -- the sorted apply pass of the shared rows -- is NOT issued; a fourth return
        value is the function that issues it.  Nothing the caller does with ``d loss/d q`` depends on that pass, so a
        caller with a second stream (``ShardedRetriever``: the query rows' exchange and update) runs the two side by side.
        ``ticket``: what ``bpr_prepare_on_owners`` has issued for this batch already (``pos`` / ``sampler`` are then unused)."""
        be, st, plan = self.backend, self.state, self.plan
        B, G = pos.numel(), plan.world
        q_gather = self._all_gather_rows_start(q)
        if ticket is None:
            pos_all, r, recv, neg_out = self._bpr_route(pos, n, sampler, want_ids)
        else:
            if ticket['B'] != B or ticket['n'] != n:
                raise ValueError('bpr_step_on_owners: the ticket was prepared for another batch shape')
            self._claim(ticket, pos, 'bpr_step_on_owners(ticket=)', sampler_id=id(sampler))
            r, recv, neg_out = ticket['route'], ticket['recv'], ticket['neg_out']
        GS, stride = G * int(getattr(be, 'BANKS', 1)), r['stride']
        q_all = q_gather()
        if ticket is not None:
            pos_rows = ticket['pos_rows']
            pos_part = be.pos_scores(self.item_local, q_all, pos_rows)
        elif hasattr(be, 'pos_rows_and_scores'):
            pos_rows, pos_part = be.pos_rows_and_scores(self.item_local, q_all, pos_all, plan, self.rank)
        else:
            pos_rows = self._own_pos_rows(pos_all)
            pos_part = be.pos_scores(self.item_local, q_all, pos_rows)
        pos_score = self._all_reduce_sum(pos_part)
        qgrad_all = torch.zeros_like(q_all)
        if ticket is not None:
            ctx = be.owner_bpr_walk(ticket['ctx'], q_all, pos_score, qgrad_all)
        else:
            ctx = be.owner_bpr_forward(st, self.item_local, q_all, recv, GS, stride, pos_rows, pos_score, n, B * G, item_grad_local,
                                       item_scale, qgrad_all, item_pad_row=0 if self.rank == 0 else -1,
                                       keys_grouped=bool(r.get('grouped', False)))
        dsum_all = self._all_reduce_sum(ctx['dsum_part'] if self._solo else ctx['dsum_part'].clone())
        if defer_rows:
            be.owner_bpr_finish(ctx, dsum_all, parts=1)
            dq = self._reduce_scatter_rows(qgrad_all, B)

            def finish_rows():
                be.owner_bpr_finish(ctx, dsum_all, parts=2)
                self._after_fixed_step()
            return ctx['loss_part'], dq, neg_out, finish_rows
        be.owner_bpr_finish(ctx, dsum_all)
        dq = self._reduce_scatter_rows(qgrad_all, B)
        self._after_fixed_step()
        return ctx['loss_part'], dq, neg_out

    def _own_pos_rows(self, pos_all):
        mine = self.plan.owner(pos_all) == self.rank
        return torch.where(mine, self.plan.local(pos_all), torch.full_like(pos_all, -1))

    def _bpr_route(self, pos, n, sampler, want_ids):
        """Steps 1 of ``bpr_step_on_owners`` that do not read a weight: the positives' ids gathered, the negatives drawn
        and routed (the positives are not), the key exchange.  -> (pos_all, route, received segments, negative ids or None)."""
        be, st, plan = self.backend, self.state, self.plan
        B = pos.numel()
        S = int(getattr(be, 'BANKS', 1))
        spec = be.sampler_spec(sampler)
        pos_all = self._all_gather_rows(pos.reshape(-1).contiguous())
        neg = None
        if spec is None:
            _, neg, _ = self.sample(sampler, B, n, pos.device, pos)
            neg = neg.contiguous()
        key = (B, n, 1, 'owners')
        cap = self._cap.get(key)
        if cap is None:
            counts = be.sample_route(st, plan, self.rank, pos, n, 1, 0, spec, self.sample_generator, neg=neg, count_only=True,
                                     banks=S, route_pos=False)
            cap = self._capacity(key[:3], int(counts.max()), store=key)
        r = be.sample_route(st, plan, self.rank, pos, n, 1, cap, spec, self.sample_generator, neg=neg, want_ids=want_ids,
                            want_logp=False, banks=S, route_pos=False, group_by_query=self.group_by_query,
                            deterministic=self.deterministic)
        recv = self._all_to_all(r['send'])
        return pos_all, r, recv, (r.get('neg_ids') if spec is not None else neg)

    def bpr_prepare_on_owners(self, pos, n, sampler, item_grad_local, item_scale=None, want_ids=False):
        """The half of ``bpr_step_on_owners`` that reads no weight and no query -- the positives' ids gathered, the negatives
        drawn and routed, the key exchange, and on the owner the sort by row, the solo classification and the queries' runs
        -- issued on the CURRENT stream; -> ticket for ``bpr_step_on_owners(..., ticket=)``.  A trainer calls it for batch
        t + 1 on a second stream before it steps batch t (``ShardedRetriever.prepare_step``): the exchange of the keys
        and a fifth of the step's kernel time then run under the previous step.  Tickets are stepped once each, in the
        order they were prepared (the draws consume the job's sample stream in that order: the negatives -- and every
        weight -- are those of the same sequence of unprepared steps)."""
        be, plan = self.backend, self.plan
        B, G = pos.numel(), plan.world
        pos_all, r, recv, neg_out = self._bpr_route(pos, n, sampler, want_ids)
        pos_rows = self._own_pos_rows(pos_all)
        GS = G * int(getattr(be, 'BANKS', 1))
        ctx = be.owner_bpr_prepare(self.state, self.item_local, B * G, recv, GS, r['stride'], pos_rows, n, B * G, item_grad_local,
                                   item_scale, item_pad_row=0 if self.rank == 0 else -1, keys_grouped=bool(r.get('grouped', False)))
        return self._stamp({'B': B, 'n': n, 'route': r, 'recv': recv, 'neg_out': neg_out, 'pos_rows': pos_rows, 'ctx': ctx},
                           pos, sampler_id=id(sampler))

    def backward(self, route, dpos, dneg, item_grad_local, item_scale=None):
        """Gradient exchange for one step.  ``route`` comes from a forward with
        ``keep_route=True``; ``dpos [B]`` / ``dneg [B, n]`` = d loss / d score on the home rank (None: the
        routed-order gradient the fused loss of the forward left in the route).  Accumulates this shard's dense item
        gradient into ``item_grad_local [rows_local, d]`` (no communication: rows never leave their owner) and returns
        d loss / d q for the own queries [B, d] (reduce-scatter of the per-owner partial sums)."""
        B = route['B']
        q_all = route['q_all']
        qgrad_all = torch.zeros_like(q_all)
        # only shard 0 holds the global padding row (item id 0), which never receives gradient
        extra = {} if item_scale is None else {'item_scale': item_scale}
        pad_row = 0 if self.rank == 0 else -1
        if 'slot_of' in route:
            G, C, stride = route['GS'], route['C'], route['stride']     # G: segments per slice (owners x banks)
            state = dict(self.state, step_dropped=route.get('dropped', self.state['step_dropped']))
            per = G * stride
            if dpos is None:
                d_send = route['d_send']
                if d_send is None:
                    raise ValueError('backward(route, None, None): the forward did not evaluate a fused loss')
            else:
                d_send = self.backend.scatter_slots(dpos.reshape(-1), dneg.reshape(B, -1), route['slot_of'], C * per)
            if C == 1:
                self.backend.backward_segments(state, self.item_local, q_all, route['recv_keys'][0], G, stride,
                                               self._all_to_all(d_send), item_grad_local, qgrad_all, item_pad_row=pad_row, **extra)
            else:
                # all gradient exchanges are issued first, the owner-side scatters follow slice by slice (slice c's
                # scatter runs while slice c+1's gradients are still on the wire)
                waits = [self._all_to_all_start(d_send[c * per:(c + 1) * per]) for c in range(C)]
                for rk, w in zip(route['recv_keys'], waits):
                    self.backend.backward_segments(state, self.item_local, q_all, rk, G, stride, w(), item_grad_local,
                                                   qgrad_all, item_pad_row=pad_row, **extra)
            return self._reduce_scatter_rows(qgrad_all, B)
        dflat = torch.cat([dpos.reshape(-1), dneg.reshape(-1)])
        d_sorted = self.backend.gather(dflat, route['positions'])
        d_owner = self._all_to_all(d_sorted, route['recv_counts'], route['send_counts'])
        self.backend.backward_keys(self.item_local, q_all, route['recv_keys'], d_owner, item_grad_local, qgrad_all,
                                   item_pad_row=pad_row, **extra)
        return self._reduce_scatter_rows(qgrad_all, B)

    def _exact_step(self, q_gather, pos, neg, keep_route):
        """Variable-split exchange: owner histogram, count exchange, host read-back, every step."""
        B, n = neg.shape
        counts = self.backend.count(pos, neg, self.plan)
        send_counts, recv_counts = self._exchange_counts(counts)
        starts = torch.tensor([0] + send_counts[:-1], dtype=torch.int64).cumsum(0)
        keys, positions = self.backend.route(pos, neg, self.plan, self.rank * B, starts)
        recv_keys = self._all_to_all(keys, recv_counts, send_counts)
        q_all = q_gather()
        scores_owner = self.backend.score_keys(self.item_local, q_all, recv_keys)
        scores_home = self._all_to_all(scores_owner, send_counts, recv_counts)
        flat = self.backend.scatter(scores_home, positions, B * (n + 1))
        out = {'pos_score': flat[:B], 'neg_score': flat[B:].view(B, n)}
        if keep_route:
            out['route'] = {'B': B, 'n': n, 'q_all': q_all, 'positions': positions, 'recv_keys': recv_keys,
                            'send_counts': send_counts, 'recv_counts': recv_counts}
        return out

    def score_ids(self, q, pos, neg, keep_route=False, q_gather=None):
        """q [B, d] own queries, pos [B], neg [B, n] GLOBAL item ids -> (pos_score [B], neg_score [B, n]).
        ``q_gather``: the wait function of an all-gather of ``q`` the caller has already started."""
        if q_gather is None:
            q_gather = self._all_gather_rows_start(q)
        if self.exchange == 'fixed':
            out = self._fixed_step(q_gather, pos, neg.shape[1], None, neg=neg.contiguous(), keep_route=keep_route)
        else:
            out = self._exact_step(q_gather, pos, neg, keep_route)
        if keep_route:
            return out['pos_score'], out['neg_score'], out['route']
        return out['pos_score'], out['neg_score']

    def sample(self, sampler, n_queries, n, device, pos):
        """(log_pos_prob, neg_ids, log_neg_prob) for this rank's queries: its rows of the job-wide draw."""
        return self.backend.sample(sampler, n_queries, n, device, pos,
                                   shard=(self.rank, self.plan.world, self.sample_generator))

    def forward_queries(self, q, pos, n, sampler, keep_route=False, fused_loss=None, mean_den=None, want_ids=True,
                        want_scores=True, want_grad=False, ticket=None):
        """BaseRetriever.forward (+ optionally the loss) for the own query vectors ``q [B, d]`` against the sharded
        table.  With one of this package's in-kernel samplers and the fixed exchange the negatives are drawn inside
        the routing launch (``neg_ids`` only with ``want_ids``); any other Sampler plugin is called and its ids are
        routed as given.  ``fused_loss`` ('bpr' | 'ssm', fixed exchange): the home kernel evaluates the loss --
        ``loss`` = sum of the row losses / ``mean_den`` (default B) -- and, with ``keep_route``, leaves d loss/d score
        in routed order in the route for ``backward(route, None, None, ...)``.  ``ticket``: what ``prepare_forward`` issued
        for this batch one step earlier (routing + key exchange)."""
        B = pos.numel()
        q_gather = self._all_gather_rows_start(q)
        spec = self.backend.sampler_spec(sampler) if self.exchange == 'fixed' else None
        if ticket is not None and spec is None:
            raise ValueError('forward_queries(ticket=...): tickets belong to the fixed exchange with an in-kernel sampler')
        if spec is not None:
            out = self._fixed_step(q_gather, pos, n, spec, keep_route=keep_route, fused_loss=fused_loss, mean_den=mean_den,
                                   want_ids=want_ids, want_scores=want_scores, want_grad=want_grad, ticket=ticket)
            if out['log_neg_prob'] is None:          # UniformSampler: int64 zeros (sampler.py:113-114), cached constants
                out['log_pos_prob'] = ops.zero_logp(pos.shape, pos.device)
                out['log_neg_prob'] = ops.zero_logp((B, n), pos.device)
        else:
            log_pos, neg, log_neg = self.sample(sampler, B, n, q.device, pos)
            if self.exchange == 'fixed':
                lp = log_pos if log_pos is not None and log_pos.is_floating_point() else None
                ln = log_neg if log_neg is not None and log_neg.is_floating_point() else None
                out = self._fixed_step(q_gather, pos, n, None, neg=neg.contiguous(), keep_route=keep_route, fused_loss=fused_loss,
                                       mean_den=mean_den, log_pos=lp, log_neg=ln, want_scores=want_scores, want_grad=want_grad)
            else:
                if fused_loss is not None:
                    raise ValueError("fused_loss needs exchange='fixed'")
                out = self._exact_step(q_gather, pos, neg, keep_route)
            out['neg_ids'], out['log_pos_prob'], out['log_neg_prob'] = neg, log_pos, log_neg
        out['query'] = q
        return out

    def sample_and_score(self, user_table, uid, pos, n, sampler, keep_route=False, **kw):
        """BaseRetriever.forward for a user-embedding query tower against the sharded item table (see
        ``forward_queries`` for the keyword arguments)."""
        return self.forward_queries(self.backend.gather_rows(user_table, uid), pos, n, sampler, keep_route=keep_route, **kw)

    # -- row look-ups for an ITEM-TOWER query encoder (SASRec: seq/sasrec.py:14, :42, :107) ----------------------------
    def lookup_rows(self, ids, keep_route=False):
        """``item_encoder(ids)`` over the sharded table for a query tower that embeds item ids itself (the history
        window of SASRec): ids out, ROWS back -- unlike the negatives, whose rows never travel, a tower needs the
        vectors.  ``ids`` (any shape, GLOBAL item ids, 0 = padding) -> ``[*ids.shape, d]`` with zero rows at the
        padding (padding ids are not sent anywhere: rank 0, which owns row 0, would be a hot owner otherwise).
        Fixed exchange (default): fixed-capacity segments, no host round trip (``_lookup_rows_fixed``); ``exchange='exact'``:
        variable split, per-owner counts exchanged and read back (one host round trip per call).  B*L*512 B per rank come
        back over xGMI either way (105 MB at B = 4096, L = 50).  Every rank must call this in lock-step."""
        be, plan, d = self.backend, self.plan, self.item_local.shape[1]
        flat = ids.reshape(-1)
        M, G = flat.numel(), plan.world
        if self._solo:
            # one rank owns every row (row 0 is the zero padding row): a plain gather, no exchange, no host round trip
            out = be.gather_rows(self.item_local, flat.contiguous()).view(*ids.shape, d)
            return (out, {'local': flat}) if keep_route else out
        if self.exchange == 'fixed' and keep_route and getattr(self, 'uniform_lookups', False):
            # a TRAINING step's look-up (its backward follows, and so does the step's routing launch, which carries a dropped
            # count to every owner) whose shape is the SAME ON EVERY RANK (``uniform_lookups``: the caller's promise -- ``fit``
            # makes it when the device-resident loader deals the rank parts of one global batch; an equal-split exchange needs
            # equal buffer sizes): fixed-capacity segments, no host round trip.  Evaluation keeps the variable split -- a full
            # segment must not hand a zero row to a forward nobody gates.
            return self._lookup_rows_fixed(ids, flat)
        sel = torch.nonzero(flat).view(-1)                         # positions that hold a real item
        vid = flat[sel]
        owner = plan.owner(vid)
        order = torch.argsort(owner, stable=True)
        send_counts, recv_counts = self._exchange_counts(torch.bincount(owner, minlength=G))
        m_send = int(sel.numel())
        recv_local = self._all_to_all(plan.local(vid)[order].contiguous(), recv_counts, send_counts)
        rows_owner = be.gather_rows(self.item_local, recv_local)
        # the rows come home in send order, behind them ONE zero row every padding position points at
        home = torch.empty(m_send + 1, d, dtype=self.item_local.dtype, device=flat.device)
        home[m_send].zero_()
        self._all_to_all(rows_owner, send_counts, recv_counts, out=home[:m_send])
        src = sel[order]                                            # position in ``flat`` of every row that came home
        inv = torch.full((M,), m_send, dtype=torch.int64, device=flat.device)
        inv[src] = torch.arange(m_send, device=flat.device)
        out = be.gather_rows(home, inv).view(*ids.shape, d)
        if keep_route:
            return out, {'src': src, 'recv_local': recv_local, 'send_counts': send_counts, 'recv_counts': recv_counts}
        return out

    def _lookup_rows_fixed(self, ids, flat):
        """The look-up over FIXED-capacity, self-describing segments (what 5.1 does for the keys): every rank sends every
        owner ``[count, dropped, C ids]`` and gets ``[C, d]`` rows back, both by EQUAL-split all-to-alls -- no count exchange,
        no ``.tolist()``, no host round trip in a step (the variable-split form above stalled the stream once per call).
        ``C`` is calibrated on the first call of a shape (the one read-back: the largest per-owner count over all ranks,
        plus slack).  A position whose owner segment is full comes back as a ZERO row and is counted: the count joins the
        step's dropped total through the next routing launch's segment headers (``state['lookup_dropped']`` ->
        ``rsa_shard_route_args.extra_dropped``), so the step is gated to a no-op and the capacity recalibrated exactly like
        an overflow of the score-side exchange.  (M = ids.numel() must be the same on every rank: ``uniform_lookups``.)"""
        be, plan, d = self.backend, self.plan, self.item_local.shape[1]
        M, G, dev = flat.numel(), plan.world, flat.device
        real = flat > 0
        owner = torch.where(real, plan.owner(flat), torch.full_like(flat, G))          # padding: bucket G, never sent
        counts = torch.bincount(owner, minlength=G + 1)                                # device
        # capacity = (the largest share of a call's positions any owner received, over all ranks, at calibration) x M + slack: a
        # function of the rank-uniform M and one all-reduced ratio, so every rank sizes its buffers alike without a collective
        # per call.  Calibration (first call, and after an overflow cleared it) is the only host read-back.
        # ``rows_share`` (``train.shard_rows_share``) fixes the share instead: 1.0 = every segment holds a whole call (no overflow
        # ever, G x the traffic); the calibrated value carries 25 % headroom and, after an overflow, the largest share seen so far.
        share_now = (counts[:G].max().to(torch.float64) / max(M, 1)).reshape(1)
        seen = self.state.get('rows_share_seen')
        seen = self.state['rows_share_seen'] = share_now if seen is None else torch.maximum(seen, share_now)    # (device, no sync)
        frac = getattr(self, 'rows_share', None) or self._cap.get(('rows_frac',))
        if frac is None:
            worst = seen.clone()
            self._all_reduce_max(worst)
            frac = self._cap[('rows_frac',)] = min(1.0, max(1.25 * float(worst.item()), 1.0 / G))
        C = max(1, min(M, (int(frac * M * self.slack) + self.margin // max(G, 1) + 255) // 256 * 256))
        order = torch.argsort(owner, stable=True)
        starts = torch.cumsum(counts, 0) - counts
        so = owner[order]
        slot_sorted = torch.arange(M, device=dev) - starts[so]
        ok = (so < G) & (slot_sorted < C)
        stride = 2 + C
        dump = G * stride                                          # one trash word behind the segments
        send = torch.full((G * stride + 1,), -1, dtype=torch.int64, device=dev)
        at = torch.where(ok, so * stride + 2 + slot_sorted, torch.full_like(so, dump))
        send[at] = plan.local(flat[order])
        kept = torch.minimum(counts[:G], torch.full_like(counts[:G], C))
        dropped = (counts[:G] - kept).sum()
        send[torch.arange(G, device=dev) * stride] = kept
        send[torch.arange(G, device=dev) * stride + 1] = dropped                      # this rank's total, in every header
        if self.state.get('lookup_dropped') is None:
            self.state['lookup_dropped'] = torch.zeros(1, dtype=torch.int32, device=dev)
        self.state['lookup_dropped'] += dropped.to(torch.int32)
        recv = self._all_to_all(send[:G * stride].contiguous()).view(G, stride)
        recv_ids = recv[:, 2:].reshape(-1)                                             # -1: dead slot
        rows_owner = be.gather_rows(self.item_local, recv_ids.clamp(min=0))
        home = torch.zeros(G * C + 1, d, dtype=self.item_local.dtype, device=dev)      # + ONE zero row for padding / dropped
        self._all_to_all(rows_owner, out=home[:G * C])
        inv = torch.full((M,), G * C, dtype=torch.int64, device=dev)
        inv[order] = torch.where(ok, so * C + slot_sorted, torch.full_like(so, G * C))
        out = be.gather_rows(home, inv).view(*ids.shape, d)
        return out, {'inv': inv, 'recv_ids': recv_ids, 'C': C}

    def lookup_rows_backward(self, route, grad, item_grad_local, scale=1.0):
        """Backward of ``lookup_rows``: the gradient rows of the real positions go to the owners of their items (the
        reverse of the forward's row exchange) and are added, times ``scale``, into ``item_grad_local`` -- this rank's
        block of the dense table gradient, or the weight block itself with ``scale = -lr`` -- by the sorted,
        atomics-free row scatter."""
        be, d = self.backend, self.item_local.shape[1]
        # an overflowed step changes no weight anywhere: the in-place form (scale = -lr into the weight block) is multiplied by
        # the step's gate (state['scale'][1] = 0 when any rank dropped an element, else 1), like the score-side update
        gate = self.state['scale'][1:2] if (item_grad_local is self.item_local and 'scale' in self.state) else None
        if 'local' in route:
            be.apply_rows(item_grad_local, route['local'], grad.reshape(-1, d).contiguous(), scale, pad_row=0, gate=gate)
            return
        if 'inv' in route:       # fixed-capacity form: the gradient rows travel back in the forward's slots (equal split)
            G, C = self.plan.world, route['C']
            g_send = torch.zeros(G * C + 1, d, dtype=grad.dtype, device=grad.device)
            g_send[route['inv']] = grad.reshape(-1, d)              # (distinct slots; padding / dropped positions: the trash row)
            g_owner = self._all_to_all(g_send[:G * C].contiguous())
            # dead slots carry id -1: the sorted scatter drops negative ids
            be.apply_rows(item_grad_local, route['recv_ids'], g_owner, scale, pad_row=0 if self.rank == 0 else -1, gate=gate)
            return
        g_send = be.gather_rows(grad.reshape(-1, d).contiguous(), route['src'])
        g_owner = self._all_to_all(g_send, route['recv_counts'], route['send_counts'])
        be.apply_rows(item_grad_local, route['recv_local'], g_owner, scale, pad_row=0 if self.rank == 0 else -1, gate=gate)

    # -- full-catalog pass (eval top-k / full softmax), sharded the same way --------------------------
    def _exchange_partials(self, x, B):
        """x [G*B, ...] (this shard's partial for EVERY query) -> [G, B, ...] (every shard's partial for the
        own queries): an equal-split all-to-all."""
        if self._solo:
            return x.view(1, B, *x.shape[1:])
        out = torch.empty_like(x)
        self.dist.all_to_all_single(out, x.contiguous(), group=self.group)
        return out.view(self.plan.world, B, *x.shape[1:])

    def full_lse_topk(self, q, k=0, want_lse=True):
        """q [B, d] own queries -> (lse [B] or None, top-k values [B, k], GLOBAL item ids [B, k]) over the
        whole sharded catalog (padding row 0 excluded), equal to the single-GPU rsa_fullscore result.
        Per shard: one MFMA pass over the local rows for all G*B gathered queries; across shards only
        (8k + 4) bytes per query per shard travel."""
        B, G = q.shape[0], self.plan.world
        has_pad = self.rank == 0
        real_rows = self.plan.n_local(self.rank) - (1 if has_pad else 0)
        q_all = self._all_gather_rows(q)
        k_local = min(int(k), real_rows)
        if real_rows <= 0 or (not want_lse and k_local == 0):
            lse_p = torch.full((G * B,), float('-inf'), dtype=torch.float32, device=q.device) if want_lse else None
            tv = ti = None
        else:
            lse_p, tv, ti = self.backend.full_partial(self.item_local, q_all, k_local, want_lse, has_pad)
        lse = None
        if want_lse:
            lse = self.backend.merge_lse(self._exchange_partials(lse_p, B).transpose(0, 1))
        if not k:
            return lse, None, None
        vals = torch.full((G * B, k), float('-inf'), dtype=torch.float32, device=q.device)
        ids = torch.zeros(G * B, k, dtype=torch.int64, device=q.device)
        if k_local:
            vals[:, :k_local] = tv
            # ti: 1-based number among the scored rows (the padding row, local row 0 of rank 0, is skipped) -> global id
            ids[:, :k_local] = self.plan.global_ids(self.rank, ti if has_pad else ti - 1)
        vals = self._exchange_partials(vals, B).transpose(0, 1).reshape(B, G * k)
        ids = self._exchange_partials(ids, B).transpose(0, 1).reshape(B, G * k)
        if self.plan.interleaved and G > 1:
            # the merge breaks ties towards the smaller COLUMN; with contiguous row blocks shard-major columns of equal value
            # are in id order, with interleaved rows they are not -- put the candidates in id order first so that equal
            # scores resolve to the smaller id on either layout, like the single-GPU kernel
            order = torch.sort(ids, dim=1, stable=True).indices
            vals, ids = torch.gather(vals, 1, order), torch.gather(ids, 1, order)
        tv, ti = self.backend.merge_topk(vals, ids, k)
        return lse, tv, ti


# ---------------------------------------------------------------------------------------------------
# Training over the sharded table
class _ShardedScoreFn(torch.autograd.Function):
    """(pos_score, neg_score) of the own queries against the sharded catalog, differentiable w.r.t. the
    queries.  The item-side gradient never leaves the owning rank: backward accumulates it straight into
    ``item_grad_local`` (this rank's [rows_local, d] block of the dense table gradient)."""

    @staticmethod
    def forward(ctx, q, table, pos, neg, item_grad_local, item_scale=None):
        pos_score, neg_score, route = table.score_ids(q, pos, neg, keep_route=True)
        ctx.table, ctx.route, ctx.item_grad_local, ctx.item_scale = table, route, item_grad_local, item_scale
        ctx.mark_non_differentiable(pos, neg)
        return pos_score, neg_score

    @staticmethod
    def backward(ctx, gpos, gneg):
        dq = ctx.table.backward(ctx.route, gpos.contiguous(), gneg.contiguous(), ctx.item_grad_local, ctx.item_scale)
        return dq, None, None, None, None, None


def sharded_scores(table, q, pos, neg, item_grad_local, item_scale=None):
    return _ShardedScoreFn.apply(q, table, pos, neg, item_grad_local, item_scale)


class _ShardedRowsFn(torch.autograd.Function):
    """``lookup_rows`` under autograd: the backward routes the gradient rows to the items' owners (collective: every
    rank runs it, at the same point of its backward graph)."""

    @staticmethod
    def forward(ctx, anchor, ids, module):
        rows, route = module.table.lookup_rows(ids, keep_route=True)
        ctx.module, ctx.route = module, route
        return rows

    @staticmethod
    def backward(ctx, g):
        m = ctx.module
        if m.grad_sink is None:
            raise RuntimeError('ShardedRows: no gradient sink bound (ShardedRows.bind(trainer)) before backward')
        m.table.lookup_rows_backward(ctx.route, g.contiguous(), m.grad_sink, m.grad_scale)
        return None, None, None


class ShardedRows(torch.nn.Module):
    """Stands in for ``item_encoder`` INSIDE a query tower when the item table is row-sharded (an ItemTowerRecommender
    such as SASRec embeds its history with the very table it scores against, seq/sasrec.py:14, :42, :107):
    ``module(ids) -> [*ids.shape, d]`` through ``ShardedItemTable.lookup_rows``, and the backward adds the rows'
    gradients into the owner's block of the table gradient (or, with in-place SGD, into the weight block) -- the
    embedding stays TIED and no rank holds, trains or all-reduces a replica of the table.  It has no parameters: the
    tower's ``parameters()`` are its dense weights only (what the bucketed all-reduce sums)."""

    def __init__(self, table):
        super().__init__()
        object.__setattr__(self, 'table', table)        # (not a submodule / buffer: the block belongs to item_encoder)
        self.grad_sink, self.grad_scale = None, 1.0
        self._anchor = None

    def bind(self, trainer):
        """Gradients go where the trainer's score exchange puts them: its gradient block, or the weights at -lr."""
        object.__setattr__(self, 'grad_sink', trainer.item_grad_local)
        self.grad_scale = 1.0 if trainer.item_sgd_lr is None else -float(trainer.item_sgd_lr)
        trainer.tower_rows = self
        return self

    @property
    def embedding_dim(self):
        return self.table.item_local.shape[1]

    def forward(self, ids):
        if not torch.is_grad_enabled():
            return self.table.lookup_rows(ids)
        if self._anchor is None or self._anchor.device != ids.device:
            # autograd only calls a Function's backward when an input requires grad; the ids cannot
            object.__setattr__(self, '_anchor', torch.zeros((), device=ids.device, requires_grad=True))
        return _ShardedRowsFn.apply(self._anchor, ids, self)


def allreduce_grads(params, dist, group=None, bucket_bytes=64 << 20):
    """Sum the dense gradients of the replicated query tower over the ranks (RCCL all-reduce) in flat
    buckets of ``bucket_bytes`` -- a few large messages instead of one per parameter (xGMI rings are
    per-link bound).  Replaces recstudio/utils/data_parallel.py:106-159, which
    re-broadcasts every parameter each step."""
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, group=group)
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        bucket, size = [], 0

    for p in params:
        if p.grad is None:
            continue
        if bucket and (bucket[0].dtype != p.grad.dtype or size + p.grad.numel() * p.grad.element_size() > bucket_bytes):
            flush()
        bucket.append(p.grad)
        size += p.grad.numel() * p.grad.element_size()
    flush()


def _record_stream_all(obj, stream):
    """``Tensor.record_stream(stream)`` on every CUDA tensor reachable through dicts / lists / tuples of ``obj``."""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream_all(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream_all(v, stream)


class ShardedRetriever:
    """Two-tower training step with the item table row-sharded over the ranks and the query tower replicated
    (data parallel).  ``query_encoder(batch_feat) -> [B, d]``; ``sampler`` / ``loss_fn`` are the usual plugins
    (Sampler / PairwiseLoss); the item block ``table.item_local`` and its gradient block are plain tensors
    owned by this rank, updated by the caller's optimizer.

    Gradients of the query tower: by default autograd fills ``query_encoder``'s ``.grad`` and the ranks sum them with
    a bucketed all-reduce, i.e. every replica ends up with the gradient a single process would compute on the
    concatenated batch.  ``sparse_query_rows`` (see ``__init__``) replaces that for a plain ``nn.Embedding`` tower."""

    def __init__(self, table, query_encoder, sampler, loss_fn, neg_count, item_sgd_lr=None, sparse_query_rows=None,
                 query_sgd_lr=None, keep_neg_ids=False, overlap_query_rows=True, owner_ssm=True):
        """``item_sgd_lr``: apply plain SGD with this learning rate to the owned item rows INSIDE the backward
        exchange (sorted scatter straight into the weight block, no [rows_local, d] gradient buffer to zero, fill
        and add); the caller then only steps the query tower.

        ``sparse_query_rows`` (opt-in; default: on exactly when ``query_sgd_lr`` is given): the replicated user
        table's gradient is row-sparse -- B rows per rank -- so the ranks exchange ``(ids, gradient rows)`` with two
        all-gathers (B*(d+2)*4 bytes per rank: 2 MB at B = 4096, d = 128) instead of all-reducing the dense
        ``[n_users, d]`` gradient (512 MB at 1 M users) that autograd would build, zero and add every step.  After a
        step ``query_rows = (ids [G*B], rows [G*B, d])`` holds the gradient of the global mean loss, identical on
        every rank, and ``query_encoder.weight.grad`` is NOT populated: the caller applies ``query_rows`` itself or
        passes ``query_sgd_lr`` (plain SGD in place; sorted, atomics-free: replicas stay bit-equal).  The path reads
        the weight rows directly, so Embedding options that act inside ``forward`` / ``backward`` are refused.

        ``keep_neg_ids``: keep the step's sampled ids in ``last_neg`` (the fused step does not write them otherwise).

        ``overlap_query_rows`` (owner-side BPR step with ``sparse_query_rows`` on a GPU): the query rows' exchange and
        update -- a dozen launches of microseconds each -- run on a second stream beside the sorted apply pass of the
        shared item rows instead of behind it (neither reads what the other writes; same results)."""
        self.table, self.query_encoder, self.sampler, self.loss_fn = table, query_encoder, sampler, loss_fn
        self.owner_ssm = bool(owner_ssm)      # stock SampledSoftmaxLoss evaluated on the owners (False: scores home, home kernel)
        self.overlap_query_rows, self._side = bool(overlap_query_rows), None
        self.neg_count = int(neg_count)
        if sparse_query_rows is None:
            sparse_query_rows = query_sgd_lr is not None
        if sparse_query_rows:
            if not isinstance(query_encoder, torch.nn.Embedding):
                raise TypeError('sparse_query_rows needs an nn.Embedding query encoder')
            if query_encoder.max_norm is not None or query_encoder.scale_grad_by_freq or query_encoder.sparse:
                raise ValueError('sparse_query_rows reads the weight rows directly: max_norm / scale_grad_by_freq / '
                                 'sparse=True are not honoured on this path')
        if query_sgd_lr is not None and not sparse_query_rows:
            raise ValueError('query_sgd_lr applies the row-sparse gradient: it needs sparse_query_rows')
        self.sparse_query_rows, self.query_sgd_lr, self.query_rows = bool(sparse_query_rows), query_sgd_lr, None
        self.keep_neg_ids, self.last_neg = bool(keep_neg_ids), None
        self.item_scale, self.item_sgd_lr, self.tower_rows = None, item_sgd_lr, None
        block = table.item_local
        for p in query_encoder.parameters() if isinstance(query_encoder, torch.nn.Module) else ():
            if p.data_ptr() == block.data_ptr() and p.numel():
                raise ValueError('the query tower holds the sharded item block as a parameter: route its item look-ups '
                                 'through shard.ShardedRows instead (BaseRetriever._setup_shard does)')
        if item_sgd_lr is None:
            self.item_grad_local = torch.zeros_like(table.item_local)
        else:
            self.item_grad_local = table.item_local
            self.item_scale = torch.full((1,), -float(item_sgd_lr), dtype=torch.float32, device=table.item_local.device)

    def set_sgd_lr(self, lr):
        """New learning rate for the in-place SGD updates (a scheduler between epochs)."""
        if self.item_scale is not None:
            self.item_scale.fill_(-float(lr))
            self.item_sgd_lr = float(lr)
            if self.tower_rows is not None:
                self.tower_rows.grad_scale = -float(lr)
        if self.query_sgd_lr is not None:
            self.query_sgd_lr = float(lr)

    def _fused_loss_kind(self):
        """'bpr' / 'ssm' when the loss can be evaluated by the home kernel of the fixed exchange (this package's stock
        BPRLoss / SampledSoftmaxLoss, exact types), else None (the loss plugin runs under autograd)."""
        from .loss_func import BPRLoss, SampledSoftmaxLoss
        if self.table.exchange != 'fixed' or not hasattr(self.table.backend, 'home'):
            return None
        if type(self.loss_fn) is BPRLoss:
            return 'bpr'
        if type(self.loss_fn) is SampledSoftmaxLoss:
            return 'ssm'
        return None

    def can_prepare(self):
        """Does ``prepare_step`` cover this trainer's step (the owner-side BPR step)?"""
        return self._fused_loss_kind() == 'bpr' and self.table.owner_loss_ok()

    def prepare_step(self, query_feat, pos_items):
        """One batch ahead: the half of the owner-side BPR step that reads no weight -- the negatives drawn and routed, the
        key exchange, the owner's sort by row / solo classification / query runs (``ShardedItemTable.bpr_prepare_on_owners``)
        -- issued on a second stream, so that it runs UNDER the step in front of it instead of at the head of its own:

            ticket = trainer.prepare_step(feat0, pos0)
            for feat1, pos1 in following_batches:
                nxt = trainer.prepare_step(feat1, pos1)            # second stream: overlaps the step below
                loss = trainer.training_step(feat0, pos0, ticket=ticket)
                feat0, pos0, ticket = feat1, pos1, nxt

        Tickets are stepped once each, in the order they were prepared; the negatives and every weight are those of the
        same sequence of ``training_step`` calls without tickets (the same work, moved in time).  Between a ``prepare_step``
        and the step of its ticket nothing else may route through this table (an evaluation pass, another trainer)."""
        table = self.table
        if not self.can_prepare():
            raise NotImplementedError('prepare_step covers the owner-side BPR step (stock BPRLoss, ShardedItemTable(owner_loss=True), '
                                      'embed_dim in the backend\'s OWNER_DIMS)')
        return table._on_second_stream(
            lambda: table.bpr_prepare_on_owners(pos_items, self.neg_count, self.sampler, self.item_grad_local, self.item_scale,
                                                want_ids=self.keep_neg_ids), pos_items)

    def training_step(self, query_feat, pos_items, label=None, ticket=None):
        """Returns this rank's share of the global mean loss (local mean / world size) after running backward:
        ``item_grad_local`` holds the gradient of the GLOBAL mean loss for the rows this rank owns (or the rows have
        been updated in place, ``item_sgd_lr``), and the query tower's gradient is the one a single process would
        compute on the concatenated batch -- in ``.grad`` (summed over ranks), or in ``query_rows`` with
        ``sparse_query_rows``."""
        table, world = self.table, self.table.plan.world
        if self.sparse_query_rows:
            weight = self.query_encoder.weight
            q = table.backend.gather_rows(weight.detach(), query_feat)
        else:
            q = self.query_encoder(query_feat)
        B = pos_items.numel()
        kind = self._fused_loss_kind()
        if ticket is not None:
            if not (kind == 'bpr' and table.owner_loss_ok()):
                raise ValueError('training_step(ticket=...): tickets belong to the owner-side BPR step')
            if ticket.get('ready') is not None:
                torch.cuda.current_stream(q.device).wait_event(ticket['ready'])
        if kind == 'bpr' and table.owner_loss_ok():
            # the loss on the owners of the negatives: one pass over the item rows per step (bpr_step_on_owners)
            if self.sparse_query_rows and self.overlap_query_rows and q.is_cuda:
                loss, dq, self.last_neg, finish_rows = table.bpr_step_on_owners(
                    q.detach(), pos_items, self.neg_count, self.sampler, self.item_grad_local, self.item_scale,
                    want_ids=self.keep_neg_ids, defer_rows=True, ticket=ticket)
                if self._side is None:
                    self._side = torch.cuda.Stream(device=q.device)
                main = torch.cuda.current_stream(q.device)
                fork = torch.cuda.Event()
                fork.record(main)
                finish_rows()                            # main stream: the shared item rows
                with torch.cuda.stream(self._side):      # side stream: the query rows (they wait for dq only)
                    self._side.wait_event(fork)
                    self._exchange_query_rows(weight, query_feat, dq)
                main.wait_stream(self._side)
                return loss
            loss, dq, self.last_neg = table.bpr_step_on_owners(q.detach(), pos_items, self.neg_count, self.sampler,
                                                               self.item_grad_local, self.item_scale, want_ids=self.keep_neg_ids,
                                                               ticket=ticket)
            if not self.sparse_query_rows and q.requires_grad:
                q.backward(dq)
        elif kind == 'ssm' and table.ssm_owner_ok() and self.owner_ssm:
            # SampledSoftmax on the owners: two phases around an 8-byte-per-query all-reduce (ssm_step_on_owners)
            loss, dq, self.last_neg = table.ssm_step_on_owners(q.detach(), pos_items, self.neg_count, self.sampler,
                                                               self.item_grad_local, self.item_scale, want_ids=self.keep_neg_ids)
            if not self.sparse_query_rows and q.requires_grad:
                q.backward(dq)
        elif kind is not None:
            # forward + loss in the exchange's own kernels; d loss/d score leaves the home kernel in routed order
            out = table.forward_queries(q.detach(), pos_items, self.neg_count, self.sampler, keep_route=True, fused_loss=kind,
                                        mean_den=B * world, want_ids=self.keep_neg_ids, want_scores=False)
            loss = out['loss']
            dq = table.backward(out['route'], None, None, self.item_grad_local, self.item_scale)
            self.last_neg = out['neg_ids']
            if not self.sparse_query_rows and q.requires_grad:
                q.backward(dq)
        else:
            if self.sparse_query_rows:
                q.requires_grad_(True)
            log_pos, neg, log_neg = table.sample(self.sampler, B, self.neg_count, q.device, pos_items)
            pos_score, neg_score = sharded_scores(table, q, pos_items, neg, self.item_grad_local, self.item_scale)
            loss = self.loss_fn(label, pos_score, log_pos, neg_score, log_neg) / world
            loss.backward()
            dq = q.grad if self.sparse_query_rows else None
            self.last_neg = neg
            loss = loss.detach()
        if self.sparse_query_rows:
            self._exchange_query_rows(weight, query_feat, dq)
        elif not table._solo:
            allreduce_grads(self.query_encoder.parameters(), table.dist, table.group)
        return loss

    def _exchange_query_rows(self, weight, query_feat, dq):
        """The row-sparse gradient of the replicated query table: (ids, rows) of every rank, applied in place when
        ``query_sgd_lr`` is set."""
        table = self.table
        ids_all = table._all_gather_rows(query_feat.reshape(-1).contiguous())
        rows_all = table._all_gather_rows(dq)
        pad = self.query_encoder.padding_idx
        self.query_rows = (ids_all, rows_all)
        if self.query_sgd_lr is not None:
            table.backend.apply_rows(weight.data, ids_all, rows_all, -float(self.query_sgd_lr),
                                     pad_row=-1 if pad is None else int(pad))

    def query_grad_dense(self):
        """The row-sparse query-table gradient of the last step as a dense ``[n_users, d]`` tensor (tests, small
        tables): what the all-reduced autograd gradient would have been."""
        ids, rows = self.query_rows
        out = torch.zeros_like(self.query_encoder.weight)
        out.index_add_(0, ids, rows)
        if self.query_encoder.padding_idx is not None:
            out[self.query_encoder.padding_idx] = 0
        return out
