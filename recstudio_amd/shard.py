"""Row-sharded item table over the GPUs of one node (BASELINE.json configs[3]).

One process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm).  Rank r owns item
rows [r*rows_per_shard, (r+1)*rows_per_shard) and B queries per step.  Per step:

  1. all_gather of the [B, d] query block                        (RCCL all-gather)
  2. sample negatives for the own queries (replicated sampler tables, own Philox stream)
  3. count + counting-sort the B*(1+n) (query, item) elements by owning rank   (HIP)
  4. all_to_all of 8-byte packed keys (query index << 32 | local row)          (RCCL all-to-all)
  5. local gather + score on the owner against the gathered queries            (HIP fused kernel)
  6. all_to_all of the fp32 scores back, scatter into [pos_score | neg_score]  (RCCL + HIP)

12 bytes per triplet cross xGMI instead of a 512-byte row.  The reference has nothing
comparable: its only multi-device mode re-broadcasts every parameter each step
(recstudio/utils/data_parallel.py:106-159) and DDP is dead code (recommender.py:731-740).

No host round trip in the step (``exchange='fixed'``, the default): every owner gets a fixed-capacity segment of
the key buffer, unused slots carry key -1 through an EQUAL-split all-to-all and are skipped by every consumer, so
the split sizes never have to be read back; the capacity comes from one calibration step (exact counts, max over
ranks, plus slack) and a sticky device-side overflow counter is checked off the critical path
(``check_overflow``).  ``exchange='exact'`` is the variable-split form (counts exchanged and read back each step).

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

import torch

from . import _native as nat
from . import ops
from . import rng
from ._native import ptr


class RowShardPlan:
    """Contiguous row blocks: owner(id) = id // rows_per_shard."""

    def __init__(self, n_items, world):
        self.n_items, self.world = int(n_items), int(world)
        self.rows_per_shard = (self.n_items + self.world - 1) // self.world

    def bounds(self, rank):
        lo = min(self.n_items, rank * self.rows_per_shard)
        return lo, min(self.n_items, lo + self.rows_per_shard)

    def owner(self, ids):
        return torch.clamp(ids // self.rows_per_shard, 0, self.world - 1)


class HipBackend:
    """Device work of the sharded step, all through the C ABI."""

    def make_generator(self, seed, device):
        return torch.Generator(device=device).manual_seed(int(seed))

    def sample(self, sampler, n_queries, n, device, pos_ids, shard=None):
        """The stand-alone Sampler plugin: (log_pos_prob, neg_ids, log_neg_prob).  ``shard = (rank, world, generator)``:
        this rank's rows of ONE global draw (rng.sharded_stream)."""
        q = torch.empty(n_queries, 1, device=device)
        if shard is None:
            return sampler(q, n, pos_ids)
        with rng.sharded_stream(*shard):
            return sampler(q, n, pos_ids)

    def new_flag(self, device):
        return torch.zeros(1, dtype=torch.int32, device=device)

    def flag_read_async(self, flag):
        """Start copying the overflow word to the host; returns poll() -> None while in flight, else the value."""
        host = torch.empty(1, dtype=torch.int32).pin_memory()
        host.copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return lambda: int(host[0]) if ev.query() else None

    def route_fixed(self, pos, neg, plan, query_base, capacity, overflow):
        B, n = neg.shape
        slots = plan.world * int(capacity)
        keys = torch.empty(slots, dtype=torch.int64, device=pos.device)
        positions = torch.empty(slots, dtype=torch.int64, device=pos.device)
        cursor = torch.empty(plan.world, dtype=torch.int32, device=pos.device)
        nat.check(nat.lib().rsa_shard_route_fixed(ptr(pos), ptr(neg), B, n, plan.rows_per_shard, plan.world, int(query_base),
                                                  int(capacity), ptr(cursor), ptr(keys), ptr(positions), ptr(overflow),
                                                  ops._stream()), 'rsa_shard_route_fixed')
        return keys, positions

    def gather_rows(self, table, ids):
        return ops.embedding_gather(table, ids)

    def count(self, pos, neg, plan):
        counts = torch.empty(plan.world, dtype=torch.int32, device=pos.device)
        n = neg.shape[1]
        nat.check(nat.lib().rsa_shard_count(ptr(pos), ptr(neg), pos.numel(), n, plan.rows_per_shard, plan.world,
                                            ptr(counts), ops._stream()), 'rsa_shard_count')
        return counts

    def route(self, pos, neg, plan, query_base, starts):
        B, n = neg.shape
        numel = B * (n + 1)
        keys = torch.empty(numel, dtype=torch.int64, device=pos.device)
        positions = torch.empty(numel, dtype=torch.int64, device=pos.device)
        cursor = starts.to(device=pos.device, dtype=torch.int32).clone()
        nat.check(nat.lib().rsa_shard_route(ptr(pos), ptr(neg), B, n, plan.rows_per_shard, plan.world, int(query_base),
                                            ptr(cursor), ptr(keys), ptr(positions), ops._stream()), 'rsa_shard_route')
        return keys, positions

    def score_keys(self, item_local, q_all, keys):
        return ops.score_packed_keys(item_local, q_all, keys)

    def scatter(self, scores, positions, numel):
        dst = torch.empty(numel, dtype=torch.float32, device=scores.device)
        nat.check(nat.lib().rsa_scatter_f32(ptr(scores), ptr(positions), scores.numel(), ptr(dst), ops._stream()),
                  'rsa_scatter_f32')
        return dst

    def gather(self, src, positions):
        dst = torch.empty(positions.numel(), dtype=torch.float32, device=src.device)
        nat.check(nat.lib().rsa_gather_f32(ptr(src), ptr(positions), positions.numel(), ptr(dst), ops._stream()),
                  'rsa_gather_f32')
        return dst

    def backward_keys(self, item_local, q_all, keys, dscore, item_grad_local, qgrad_all, item_pad_row=-1, item_scale=None):
        """Owner side of the backward: item_grad_local[row] += d * q_all[qidx]; qgrad_all[qidx] += d * item_local[row].
        ``item_scale`` (device scalar): multiply the item-side update by it -- with ``item_grad_local is item_local``
        and item_scale = -lr this is plain SGD applied in place (the query-side pass, which reads the rows, runs
        first)."""
        m = keys.numel()
        if m == 0:
            return
        rows = torch.empty(m, dtype=torch.int64, device=keys.device)
        qidx = torch.empty(m, dtype=torch.int64, device=keys.device)
        nat.check(nat.lib().rsa_shard_unpack(ptr(keys), m, ptr(rows), ptr(qidx), ops._stream()), 'rsa_shard_unpack')
        if item_local.shape[1] in (64, 128, 256):
            # two atomics-free sorted scatters (by item row, then by query): reproducible, ~2x faster than atomics
            ops.scatter_rows_sorted(qgrad_all, item_local, qidx.view(m, 1), dscore.view(m, 1), query_index=rows, pad_row=-1)
            ops.scatter_rows_sorted(item_grad_local, q_all, rows.view(m, 1), dscore.view(m, 1), query_index=qidx,
                                    upstream=item_scale, pad_row=item_pad_row)
            return
        if item_scale is not None:
            raise NotImplementedError('in-place item update needs embed_dim in {64, 128, 256}')
        rows, qidx = rows.clamp_(min=0), qidx.clamp_(min=0)      # empty slots (d = 0 there): any valid row
        ops.fused_backward(item_local, q_all, rows.view(m, 1), dscore.view(m, 1), query_index=qidx,
                           dense_item_grad=False, want_query_grad=False, item_grad_out=item_grad_local,
                           query_table_grad=qgrad_all, query_table_pad_row=-1, item_pad_row=item_pad_row)


    def apply_rows(self, table, ids, rows, scale, pad_row=0):
        """table[ids[e]] += scale * rows[e], duplicates summed in sorted order without atomics: every replica that
        applies the same (ids, rows) ends up with the same bits."""
        m = ids.numel()
        if m == 0:
            return
        d = rows.shape[1]
        if d in (64, 128, 256):
            ops.scatter_rows_sorted(table, rows, ids.view(m, 1), torch.full((m, 1), float(scale), device=rows.device),
                                    query_index=torch.arange(m, device=ids.device), pad_row=pad_row)
        else:
            keep = (ids != pad_row).to(rows.dtype).unsqueeze(1) * float(scale)     # the padding row takes no gradient
            ops.scatter_add_rows(rows * keep, ids, table.shape[0], out=table)

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
        """vals/ids [B, G*k] (shard-major, each shard's list sorted) -> exact global top-k, ties -> smaller id."""
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


class ShardedItemTable:
    def __init__(self, item_local, plan, rank, dist, backend=None, group=None, exchange='fixed', slack=1.08,
                 margin=4096, check_every=64, sample_seed=2022, chunks=1):
        """``chunks`` > 1 (fixed-capacity exchange only): the step's queries are cut into that many contiguous
        slices whose exchanges are issued asynchronously, so that slice c+1's key all-to-all and slice c-1's score
        all-to-all travel over xGMI while slice c is being scored (see ``_score_ids_pipelined``)."""
        self.item_local, self.plan, self.rank, self.dist = item_local, plan, int(rank), dist
        self.chunks = max(1, int(chunks))
        self.backend = backend if backend is not None else HipBackend()
        self.group = group
        if exchange not in ('fixed', 'exact'):
            raise ValueError("exchange must be 'fixed' or 'exact'")
        self.exchange, self.slack, self.margin, self.check_every = exchange, float(slack), int(margin), int(check_every)
        self._cap, self._steps, self._poll, self._poll_due = {}, 0, None, 0
        self._overflow = self.backend.new_flag(item_local.device)
        # ONE sampler stream for the whole job: same seed on every rank, advanced in lock-step (see module docstring)
        self.sample_generator = self.backend.make_generator(sample_seed, item_local.device)
        # the query all-gather runs on its OWN communicator: collectives of one communicator execute in issue
        # order on one stream, so on the main group the key / score exchanges would queue behind the (at n = 64)
        # much larger gather instead of overlapping with it
        self.gather_group = _gather_group(dist) if (group is None and plan.world > 1) else group
        lo, hi = plan.bounds(rank)
        if item_local.shape[0] != hi - lo:
            raise ValueError(f'rank {rank} must hold rows [{lo}, {hi}) of the item table, got {item_local.shape[0]}')

    # -- collectives (RCCL through torch.distributed) -------------------------------------------
    def _all_gather_rows(self, x):
        return self._all_gather_rows_start(x)()

    def _all_gather_rows_start(self, x):
        """Start the all-gather of the query block on the communicator's own stream and return a function
        that waits for it.  At B = 65536 queries/GPU the block is 33.5 MB per rank -- the largest message of
        the step (7 x 33.5 MB arrive per GPU over xGMI, vs 8 + 4 bytes per triplet for keys and scores) -- and
        nothing before the owner-side scoring needs it, so it flies under sampling and routing."""
        out = torch.empty(self.plan.world * x.shape[0], *x.shape[1:], dtype=x.dtype, device=x.device)
        x = x.contiguous()
        work = self.dist.all_gather_into_tensor(out, x, group=self.gather_group, async_op=True)

        def wait():
            work.wait()
            return out
        wait.keep = x           # the source must stay alive until the collective has run
        return wait

    def _exchange_counts(self, counts):
        send = counts.to(torch.int64)
        recv = torch.empty_like(send)
        self.dist.all_to_all_single(recv, send, group=self.group)
        return [int(v) for v in send.tolist()], [int(v) for v in recv.tolist()]

    def _all_to_all(self, x, recv_counts=None, send_counts=None):
        """Variable split (host lists) or, with no counts, the equal split of the fixed-capacity exchange."""
        if recv_counts is None:
            out = torch.empty_like(x)
            self.dist.all_to_all_single(out, x, group=self.group)
            return out
        out = torch.empty(sum(recv_counts), dtype=x.dtype, device=x.device)
        self.dist.all_to_all_single(out, x, output_split_sizes=recv_counts, input_split_sizes=send_counts,
                                    group=self.group)
        return out

    def _all_to_all_start(self, x):
        """Equal-split all-to-all issued asynchronously (it runs on the communicator's stream once everything queued
        on the current stream so far has finished); returns wait() -> the received tensor, ordered after the transfer
        on the current stream."""
        out = torch.empty_like(x)
        work = self.dist.all_to_all_single(out, x, group=self.group, async_op=True)

        def wait():
            if work is not None:
                work.wait()
            return out
        wait.keep = x
        return wait

    def _reduce_scatter_rows(self, x, rows_per_rank):
        out = torch.empty(rows_per_rank, *x.shape[1:], dtype=x.dtype, device=x.device)
        self.dist.reduce_scatter_tensor(out, x.contiguous(), group=self.group)
        return out

    # -- fixed-capacity bookkeeping -------------------------------------------------------------------
    def _calibrate(self, key, send_counts):
        """Capacity of one owner segment from an exact step: the largest per-owner count of ANY rank (so that every
        rank uses the same equal split) plus slack.  One small all-reduce + read-back, on the calibration step only."""
        B, n = key
        m = torch.tensor([max(send_counts)], dtype=torch.int64, device=self.item_local.device)
        self.dist.all_reduce(m, op=self.dist.ReduceOp.MAX, group=self.group)
        cap = int(int(m.item()) * self.slack) + self.margin
        cap = min(B * (n + 1), (cap + 255) // 256 * 256)
        self._cap[key] = max(cap, 1)

    def _start_overflow_read(self):
        """The overflow words of ALL ranks summed (one tiny all-reduce) and copied to the host asynchronously: every
        rank sees the same value, so every rank raises at the same step -- a rank that raised alone would leave the
        others waiting in their next collective."""
        total = self._overflow.clone()
        self.dist.all_reduce(total, group=self.group)
        return self.backend.flag_read_async(total)

    def check_overflow(self, block=True):
        """Raise if any routed element of ANY rank ever found its owner's segment full (its score was lost).
        COLLECTIVE (call it on every rank at the same point)."""
        self._poll = self._poll or self._start_overflow_read()
        v = self._poll()
        while v is None:
            v = self._poll()
        self._poll = None
        if v:
            self._cap.clear()               # recalibrate on the next step
            self._overflow.zero_()
            raise RuntimeError(f'sharded exchange: {v} elements (all ranks) did not fit their owner segment (capacity slack '
                               f'{self.slack}); the affected steps are invalid -- raise `slack` / `margin` or use '
                               "exchange='exact' for id distributions that drift this fast")

    LAG = 8          # steps between starting the read-back of the overflow words and looking at it

    def _after_fixed_step(self):
        """Every ``check_every`` steps the (all-reduced) overflow count starts its way to the host; LAG steps later --
        long after the copy has landed, so without a stall -- every rank looks at it, at the SAME step."""
        self._steps += 1
        if self.check_every <= 0:
            return
        if self._poll is not None and self._steps >= self._poll_due:
            self.check_overflow()
        if self._poll is None and self._steps % self.check_every == 0:
            self._poll = self._start_overflow_read()
            self._poll_due = self._steps + self.LAG

    # -- the step ---------------------------------------------------------------------------------
    def backward(self, route, dpos, dneg, item_grad_local, item_scale=None):
        """Gradient exchange for one step.  ``route`` comes from
        ``score_ids(..., keep_route=True)``; ``dpos [B]`` / ``dneg [B, n]`` = d loss / d score on the home
        rank.  Accumulates this shard's dense item gradient into ``item_grad_local [rows_local, d]`` (no
        communication: rows never leave their owner) and returns d loss / d q for the own queries [B, d]
        (reduce-scatter of the per-owner partial sums)."""
        B = route['B']
        q_all = route['q_all']
        qgrad_all = torch.zeros_like(q_all)
        # only shard 0 holds the global padding row (item id 0), which never receives gradient
        extra = {} if item_scale is None else {'item_scale': item_scale}
        pad_row = 0 if self.rank == 0 else -1
        if 'slices' in route:
            # pipelined step: all gradient exchanges are issued first, the owner-side scatters follow slice by slice
            # (slice c's scatter runs while slice c+1's gradients are still on the wire)
            dpos, dneg = dpos.reshape(-1), dneg.reshape(B, -1)
            waits = []
            for (b0, b1, positions, _) in route['slices']:
                dflat = torch.cat([dpos[b0:b1], dneg[b0:b1].reshape(-1)])
                waits.append(self._all_to_all_start(self.backend.gather(dflat, positions)))
            for (_, _, _, recv_keys), wait in zip(route['slices'], waits):
                self.backend.backward_keys(self.item_local, q_all, recv_keys, wait(), item_grad_local, qgrad_all,
                                           item_pad_row=pad_row, **extra)
            return self._reduce_scatter_rows(qgrad_all, B)
        dflat = torch.cat([dpos.reshape(-1), dneg.reshape(-1)])
        d_sorted = self.backend.gather(dflat, route['positions'])           # empty slots: 0
        d_owner = self._all_to_all(d_sorted, route.get('recv_counts'), route.get('send_counts'))
        self.backend.backward_keys(self.item_local, q_all, route['recv_keys'], d_owner, item_grad_local, qgrad_all,
                                   item_pad_row=pad_row, **extra)
        return self._reduce_scatter_rows(qgrad_all, B)

    def _score_ids_pipelined(self, q_gather, pos, neg, keep_route):
        """The fixed-capacity step cut into ``self.chunks`` query slices.  Issue order on the current stream:
        route(0), route(1), ... (each followed by its asynchronous key all-to-all on the communicator's stream),
        then per slice: wait for its keys, score, start the score all-to-all; finally per slice: wait, scatter home.
        Collectives of one communicator run in issue order, so slice c's keys arrive while slice c-1 is being scored
        and its scores go back while slice c+1 is being scored.  Every rank issues the same sequence."""
        B, n = neg.shape
        C = self.chunks
        Bc = B // C
        cap = self._cap[(Bc, n)]
        bounds = [(c * Bc, (c + 1) * Bc) for c in range(C)]
        routed = []
        for b0, b1 in bounds:
            keys, positions = self.backend.route_fixed(pos[b0:b1], neg[b0:b1], self.plan, self.rank * B + b0, cap,
                                                       self._overflow)
            routed.append((positions, self._all_to_all_start(keys)))
        q_all = q_gather()
        scored = []
        for positions, wait_keys in routed:
            recv_keys = wait_keys()
            scored.append((positions, recv_keys,
                           self._all_to_all_start(self.backend.score_keys(self.item_local, q_all, recv_keys))))
        pos_parts, neg_parts, slices = [], [], []
        for (b0, b1), (positions, recv_keys, wait_scores) in zip(bounds, scored):
            flat = self.backend.scatter(wait_scores(), positions, Bc * (n + 1))
            pos_parts.append(flat[:Bc])
            neg_parts.append(flat[Bc:].view(Bc, n))
            slices.append((b0, b1, positions, recv_keys))
        self._after_fixed_step()
        pos_score, neg_score = torch.cat(pos_parts), torch.cat(neg_parts)
        if keep_route:
            return pos_score, neg_score, {'B': B, 'n': n, 'q_all': q_all, 'slices': slices}
        return pos_score, neg_score

    def score_ids(self, q, pos, neg, keep_route=False, q_gather=None):
        """q [B, d] own queries, pos [B], neg [B, n] GLOBAL item ids -> (pos_score [B], neg_score [B, n]).
        ``q_gather``: the wait function of an all-gather of ``q`` the caller has already started."""
        B, n = neg.shape
        plan = self.plan
        if q_gather is None:
            q_gather = self._all_gather_rows_start(q)
        if self.exchange == 'fixed' and self.chunks > 1 and B % self.chunks == 0:
            Bc = B // self.chunks
            if (Bc, n) not in self._cap:
                # capacity of a slice's owner segments: exact owner counts of slice 0, max over ranks, plus slack
                send_counts, _ = self._exchange_counts(self.backend.count(pos[:Bc], neg[:Bc], plan))
                self._calibrate((Bc, n), send_counts)
            return self._score_ids_pipelined(q_gather, pos, neg, keep_route)
        cap = self._cap.get((B, n)) if self.exchange == 'fixed' else None
        route = {'B': B, 'n': n}
        if cap is None:
            # exact split sizes: owner histogram, count exchange, host read-back (every step with exchange='exact',
            # the calibration step otherwise)
            counts = self.backend.count(pos, neg, plan)
            send_counts, recv_counts = self._exchange_counts(counts)
            starts = torch.tensor([0] + send_counts[:-1], dtype=torch.int64).cumsum(0)
            keys, positions = self.backend.route(pos, neg, plan, self.rank * B, starts)
            recv_keys = self._all_to_all(keys, recv_counts, send_counts)
            route.update(send_counts=send_counts, recv_counts=recv_counts)
            if self.exchange == 'fixed':
                self._calibrate((B, n), send_counts)
            back = (send_counts, recv_counts)
        else:
            keys, positions = self.backend.route_fixed(pos, neg, plan, self.rank * B, cap, self._overflow)
            recv_keys = self._all_to_all(keys)
            back = (None, None)
        q_all = q_gather()
        scores_owner = self.backend.score_keys(self.item_local, q_all, recv_keys)
        scores_home = self._all_to_all(scores_owner, *back)
        flat = self.backend.scatter(scores_home, positions, B * (n + 1))
        if cap is not None:
            self._after_fixed_step()
        if keep_route:
            route.update(q_all=q_all, positions=positions, recv_keys=recv_keys)
            return flat[:B], flat[B:].view(B, n), route
        return flat[:B], flat[B:].view(B, n)

    def sample(self, sampler, n_queries, n, device, pos):
        """(log_pos_prob, neg_ids, log_neg_prob) for this rank's queries: its rows of the job-wide draw."""
        return self.backend.sample(sampler, n_queries, n, device, pos,
                                   shard=(self.rank, self.plan.world, self.sample_generator))

    def sample_and_score(self, user_table, uid, pos, n, sampler, keep_route=False):
        """BaseRetriever.forward for a user-embedding query tower against the sharded item table."""
        B = uid.numel()
        q = self.backend.gather_rows(user_table, uid)
        q_gather = self._all_gather_rows_start(q)
        log_pos, neg, log_neg = self.sample(sampler, B, n, uid.device, pos)
        res = self.score_ids(q, pos, neg, keep_route, q_gather)
        out = {'pos_score': res[0], 'neg_score': res[1], 'neg_ids': neg, 'log_pos_prob': log_pos,
               'log_neg_prob': log_neg, 'query': q}
        if keep_route:
            out['route'] = res[2]
        return out

    # -- full-catalog pass (eval top-k / full softmax), sharded the same way --------------------------
    def _exchange_partials(self, x, B):
        """x [G*B, ...] (this shard's partial for EVERY query) -> [G, B, ...] (every shard's partial for the
        own queries): an equal-split all-to-all."""
        out = torch.empty_like(x)
        self.dist.all_to_all_single(out, x.contiguous(), group=self.group)
        return out.view(self.plan.world, B, *x.shape[1:])

    def full_lse_topk(self, q, k=0, want_lse=True):
        """q [B, d] own queries -> (lse [B] or None, top-k values [B, k], GLOBAL item ids [B, k]) over the
        whole sharded catalog (padding row 0 excluded), equal to the single-GPU rsa_fullscore result.
        Per shard: one MFMA pass over the local rows for all G*B gathered queries; across shards only
        (8k + 4) bytes per query per shard travel."""
        B, G = q.shape[0], self.plan.world
        lo, hi = self.plan.bounds(self.rank)
        has_pad = self.rank == 0
        real_rows = (hi - lo) - (1 if has_pad else 0)
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
            ids[:, :k_local] = ti + (lo if has_pad else lo - 1)          # local 1-based row number -> global id
        vals = self._exchange_partials(vals, B).transpose(0, 1).reshape(B, G * k)
        ids = self._exchange_partials(ids, B).transpose(0, 1).reshape(B, G * k)
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


class ShardedRetriever:
    """Two-tower training step with the item table row-sharded over the ranks and the query tower replicated
    (data parallel).  ``query_encoder(batch_feat) -> [B, d]``; ``sampler`` / ``loss_fn`` are the usual plugins
    (Sampler / PairwiseLoss); the item block ``table.item_local`` and its gradient block are plain tensors
    owned by this rank, updated by the caller's optimizer."""

    def __init__(self, table, query_encoder, sampler, loss_fn, neg_count, item_sgd_lr=None, sparse_query_rows=None,
                 query_sgd_lr=None):
        """``item_sgd_lr``: apply plain SGD with this learning rate to the owned item rows INSIDE the backward
        exchange (sorted scatter straight into the weight block, no [rows_local, d] gradient buffer to zero, fill
        and add); the caller then only steps the query tower.

        ``sparse_query_rows`` (default: True when ``query_encoder`` is a plain ``nn.Embedding``): the replicated
        user table's gradient is row-sparse -- B rows per rank -- so the ranks exchange ``(ids, gradient rows)`` with
        two all-gathers (B*(d+2)*4 bytes per rank: 2 MB at B = 4096, d = 128) instead of all-reducing the dense
        ``[n_users, d]`` gradient (512 MB at 1 M users) that autograd would build, zero and add every step.  After a
        step ``query_rows = (ids [G*B], rows [G*B, d])`` holds the gradient of the global mean loss, identical on
        every rank; ``query_sgd_lr`` applies plain SGD with it in place (sorted, atomics-free: replicas stay bit-equal)."""
        self.table, self.query_encoder, self.sampler, self.loss_fn = table, query_encoder, sampler, loss_fn
        self.neg_count = int(neg_count)
        if sparse_query_rows is None:
            sparse_query_rows = type(query_encoder) is torch.nn.Embedding
        if sparse_query_rows and not isinstance(query_encoder, torch.nn.Embedding):
            raise TypeError('sparse_query_rows needs an nn.Embedding query encoder')
        if query_sgd_lr is not None and not sparse_query_rows:
            raise ValueError('query_sgd_lr applies the row-sparse gradient: it needs sparse_query_rows')
        self.sparse_query_rows, self.query_sgd_lr, self.query_rows = bool(sparse_query_rows), query_sgd_lr, None
        self.item_scale = None
        if item_sgd_lr is None:
            self.item_grad_local = torch.zeros_like(table.item_local)
        else:
            self.item_grad_local = table.item_local
            self.item_scale = torch.full((1,), -float(item_sgd_lr), dtype=torch.float32, device=table.item_local.device)

    def training_step(self, query_feat, pos_items, label=None):
        """Returns this rank's share of the global mean loss (local mean / world size) after running backward:
        ``item_grad_local`` holds the gradient of the GLOBAL mean loss for the rows this rank owns, and the
        query tower's ``.grad`` is summed over ranks, i.e. every replica ends up with the same gradient a
        single process would compute on the concatenated batch."""
        table, world = self.table, self.table.plan.world
        if self.sparse_query_rows:
            weight = self.query_encoder.weight
            q = table.backend.gather_rows(weight.detach(), query_feat).requires_grad_(True)
        else:
            q = self.query_encoder(query_feat)
        B = pos_items.numel()
        log_pos, neg, log_neg = table.sample(self.sampler, B, self.neg_count, q.device, pos_items)
        pos_score, neg_score = sharded_scores(table, q, pos_items, neg, self.item_grad_local, self.item_scale)
        loss = self.loss_fn(label, pos_score, log_pos, neg_score, log_neg) / world
        loss.backward()
        if self.sparse_query_rows:
            ids_all = table._all_gather_rows(query_feat.reshape(-1).contiguous())
            rows_all = table._all_gather_rows(q.grad)
            pad = self.query_encoder.padding_idx
            self.query_rows = (ids_all, rows_all)
            if self.query_sgd_lr is not None:
                table.backend.apply_rows(weight.data, ids_all, rows_all, -float(self.query_sgd_lr),
                                         pad_row=-1 if pad is None else int(pad))
        else:
            allreduce_grads(self.query_encoder.parameters(), table.dist, table.group)
        self.last_neg = neg
        return loss.detach()

    def query_grad_dense(self):
        """The row-sparse query-table gradient of the last step as a dense ``[n_users, d]`` tensor (tests, small
        tables): what the all-reduced autograd gradient would have been."""
        ids, rows = self.query_rows
        out = torch.zeros_like(self.query_encoder.weight)
        out.index_add_(0, ids, rows)
        if self.query_encoder.padding_idx is not None:
            out[self.query_encoder.padding_idx] = 0
        return out
