"""CPU, world_size = 2, gloo: the row-sharded exchange protocol (split sizes, all-to-all order,
reassembly) must reproduce single-process scores exactly.  The device work is supplied by a
checker backend built on the oracle (this test is the importer; the product has no CPU path)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from oracle import philox

# the spawned ranks are fresh interpreters: a few OpenMP threads each instead of one per core per rank (8 ranks x all
# cores made the world-8 cases crawl)
os.environ.setdefault('OMP_NUM_THREADS', '2')


class CheckerBackend:
    """Same interface as recstudio_amd.shard.HipBackend, restated with torch-CPU ops."""

    def gather_rows(self, table, ids):
        return table[ids]

    # the device generator's role is played by (seed, offset) of the oracle's Philox restatement: ONE stream for the
    # job, rank r owns rows [r*B, (r+1)*B) of the global [G*B, n] draw
    GRID = 256 * 1024

    def make_generator(self, seed, device):
        return {'seed': int(seed), 'offset': 0}

    def sample(self, sampler, n_queries, n, device, pos_ids, shard=None):
        if shard is None:
            return sampler.forward(torch.zeros(n_queries, 1), n, pos_ids)
        rank, world, gen = shard
        numel = n_queries * world * n
        grid = philox.rng_grid_threads(numel, 256, 2048)
        res = sampler.forward_device_stream(torch.zeros(n_queries * world, 1), n, gen['seed'], gen['offset'], grid,
                                            pos_items=None if pos_ids is None else pos_ids.repeat(world))
        lp, ids, lnp = res if pos_ids is not None else (None, *res)
        unroll = 4 if isinstance(sampler, oracle.PopularSamplerModel) or sampler.num_items < (1 << 28) else 2
        gen['offset'] += ((numel - 1) // (grid * unroll) + 1) * 4
        rows = slice(rank * n_queries, (rank + 1) * n_queries)
        if pos_ids is None:
            return ids[rows], lnp[rows]
        return lp[rows], ids[rows], lnp[rows]

    def flag_read_async(self, flag):
        return lambda: int(flag[0])

    # -- version 2 of the fixed-capacity exchange, restated with torch-CPU ops --------------------------------------
    HDR = 2

    def new_state(self, device):
        return {'overflow': torch.zeros(1, dtype=torch.int32), 'step_dropped': torch.zeros(1, dtype=torch.int32),
                'scale': torch.ones(2)}

    def sampler_spec(self, sampler):
        """The oracle's two samplers play the role of the in-kernel ones."""
        if isinstance(sampler, (oracle.UniformSampler, oracle.PopularSamplerModel)):
            return {'sampler': sampler}
        return None

    def sample_route(self, state, plan, rank, pos, n, chunks, capacity, spec, generator, neg=None, want_ids=False,
                     want_logp=False, count_only=False, banks=1, route_pos=True, group_by_query=False, deterministic=False):
        assert banks == 1                                      # (the checker has no BANKS attribute: one segment per owner)
        B, G, C = pos.numel(), plan.world, chunks
        out = {}
        if spec is not None:
            off0 = generator['offset']
            lp, neg, lnp = self.sample(spec['sampler'], B, n, None, pos, shard=(rank, G, generator))
            if count_only:
                generator['offset'] = off0
            if want_ids:
                out['neg_ids'] = neg
            if want_logp and isinstance(spec['sampler'], oracle.PopularSamplerModel):
                out['log_pos_prob'], out['log_neg_prob'] = lp, lnp
        else:
            out['neg_ids'] = neg
        ids = self._elements(pos, neg)                         # [B * (1 + n)], query-major, column 0 = positive
        owner = plan.owner(ids)
        m = torch.arange(B).repeat_interleave(n + 1)
        Bc = B // C
        sl = m // Bc
        routed = torch.ones(B, n + 1, dtype=torch.bool)
        if not route_pos:
            routed[:, 0] = False                               # owner-side BPR step: the positives do not travel
        routed = routed.reshape(-1)
        if count_only:
            return torch.bincount((sl * G + owner)[routed], minlength=C * G).to(torch.int32)
        stride = capacity + self.HDR
        send = torch.full((C * G * stride,), -7, dtype=torch.int64)           # slack: never read by a consumer
        slot_of = torch.full((B * (n + 1),), -1, dtype=torch.int32)
        key = ((rank * B + m) << 32) | plan.local(ids)
        dropped = 0
        for seg in range(C * G):
            sel = torch.nonzero((sl * G + owner == seg) & routed).flatten()
            dropped += max(0, sel.numel() - capacity)
            sel = sel[:capacity]
            base = seg * stride
            send[base] = sel.numel()
            send[base + self.HDR:base + self.HDR + sel.numel()] = key[sel]
            slot_of[sel] = (base + self.HDR + torch.arange(sel.numel())).to(torch.int32)
        if state.get('lookup_dropped') is not None:            # drops of the step's tower look-ups: gated like a routing overflow
            dropped += int(state['lookup_dropped'])
            state['lookup_dropped'].zero_()
        send[1::stride][:C * G] = dropped
        out.update(send=send, slot_of=slot_of, stride=stride)
        return out

    def _live(self, keys, n_seg, stride):
        k = keys.view(n_seg, stride)
        within = torch.arange(stride).view(1, -1)
        return ((within >= self.HDR) & (within - self.HDR < k[:, :1])).reshape(-1)

    def score_segments(self, state, item_local, q_all, recv_keys, n_seg, stride, first=True, out=None):
        live = self._live(recv_keys, n_seg, stride)
        keys = torch.where(live, recv_keys, torch.zeros((), dtype=torch.int64))
        rows, qidx = keys & 0xffffffff, keys >> 32
        sc = torch.where(live, (item_local[rows] * q_all[qidx]).sum(-1), torch.full((), float('nan')))   # slack: garbage
        if first:
            total = int(recv_keys.view(n_seg, stride)[:, 1].sum())
            state['step_dropped'][0] = total
            state['overflow'] += total
        if out is not None:
            out.copy_(sc)
            return out
        return sc

    def home(self, scores_home, slot_of, B, n, loss=None, pos_logp=None, neg_logp=None, mean_den=None,
             want_scores=True, want_grad=False, want_dsend=False):
        so = slot_of.long().view(B, n + 1)
        live = so >= 0
        sc = torch.where(live, scores_home[so.clamp(min=0)], torch.zeros(()))
        pos, neg = sc[:, 0], sc[:, 1:]
        out = {}
        if want_scores or loss is None:
            out['pos_score'], out['neg_score'] = pos.clone(), neg.clone()
        if loss is None:
            return out
        den = float(mean_den if mean_den is not None else B)
        pos_r, neg_r = pos.clone().requires_grad_(True), neg.clone().requires_grad_(True)
        lp = live[:, :1] & live[:, 1:]                          # a term needs its negative AND the query's positive
        if loss == 'bpr':
            rows = -(torch.where(lp, torch.nn.functional.logsigmoid(pos_r.view(-1, 1) - neg_r), torch.zeros(()))).sum(-1) / n
        else:
            zp = pos_r - (pos_logp if pos_logp is not None else 0)
            zn = neg_r - (neg_logp if neg_logp is not None else 0)
            zn = torch.where(live[:, 1:], zn, torch.full((), float('-inf')))
            rows = torch.logsumexp(torch.cat([zp.view(-1, 1), zn], 1), -1) - zp
            rows = torch.where(live[:, 0], rows, torch.zeros(()))
        total = rows.sum() / den
        total.backward()
        out['loss'], out['row_loss'] = total.detach(), rows.detach()
        dpos, dneg = pos_r.grad, neg_r.grad
        if want_grad:
            out['dpos'], out['dneg'] = dpos, dneg
        if want_dsend:
            out['d_send'] = self.scatter_slots(dpos, dneg, slot_of, scores_home.numel())
        return out

    def scatter_slots(self, dpos, dneg, slot_of, n_slots):
        B = dpos.numel()
        d = torch.cat([dpos.view(B, 1), dneg.view(B, -1)], 1).reshape(-1)
        d_send = torch.full((n_slots,), float('nan'))                         # slack: garbage, never read
        live = slot_of >= 0
        d_send[slot_of[live].long()] = d[live]
        return d_send

    def backward_segments(self, state, item_local, q_all, recv_keys, n_seg, stride, d_owner, item_grad_local, qgrad_all,
                          item_pad_row=-1, item_scale=None):
        gate = 0.0 if int(state['step_dropped'][0]) else 1.0
        self._publish_scale(state, gate, item_scale)
        keep = self._live(recv_keys, n_seg, stride)
        keys, dscore = recv_keys[keep], d_owner[keep]
        rows, qidx = keys & 0xffffffff, keys >> 32
        live = rows != item_pad_row
        qgrad_all.index_add_(0, qidx, gate * dscore.unsqueeze(1) * item_local[rows])       # reads the rows: first
        scale = gate * (1.0 if item_scale is None else float(item_scale))
        item_grad_local.index_add_(0, rows[live], scale * dscore[live].unsqueeze(1) * q_all[qidx[live]])

    @staticmethod
    def _publish_scale(state, gate, item_scale):
        """what the owner kernels leave in scale_out: {gate * item_scale, gate} (read by the tower-side tied update)"""
        state['scale'][0] = gate * (1.0 if item_scale is None else float(item_scale))
        state['scale'][1] = gate

    # -- the stock BPR step evaluated on the owners, restated with torch-CPU ops -----------------------------------
    OWNER_DIMS = range(1, 4097)

    def pos_scores(self, item_local, q_all, pos_rows):
        own = pos_rows >= 0
        return torch.where(own, (item_local[pos_rows.clamp(min=0)] * q_all).sum(-1), torch.zeros(()))

    def owner_bpr_forward(self, state, item_local, q_all, recv_keys, n_seg, stride, pos_rows, pos_score, n, mean_den,
                          item_target, item_scale, qgrad_all, item_pad_row=-1, keys_grouped=False):
        total = int(recv_keys.view(n_seg, stride)[:, 1].sum())
        state['step_dropped'][0] = total
        state['overflow'] += total
        gate = 0.0 if total else 1.0
        self._publish_scale(state, gate, item_scale)
        keys = recv_keys[self._live(recv_keys, n_seg, stride)]
        rows, qidx = keys & 0xffffffff, keys >> 32
        x = pos_score[qidx] - (item_local[rows] * q_all[qidx]).sum(-1)              # loss_func.py:55-59, one term per element
        d = torch.sigmoid(-x) / (n * mean_den)
        loss_part = (-torch.nn.functional.logsigmoid(x) / n).sum() / mean_den
        qgrad_all.index_add_(0, qidx, gate * d.unsqueeze(1) * item_local[rows])
        dsum_part = torch.zeros(q_all.shape[0]).index_add_(0, qidx, d)
        return {'rows': rows, 'qidx': qidx, 'd': d, 'gate': gate, 'item_local': item_local, 'q_all': q_all, 'pos_rows': pos_rows,
                'item_target': item_target, 'item_scale': item_scale, 'qgrad_all': qgrad_all, 'pad': item_pad_row,
                'dsum_part': dsum_part, 'loss_part': loss_part}

    def owner_bpr_prepare(self, state, item_local, Q, recv_keys, n_seg, stride, pos_rows, n, mean_den, item_target, item_scale,
                          item_pad_row=-1, keys_grouped=False):
        return {'prepared': (state, item_local, recv_keys, n_seg, stride, pos_rows, n, mean_den, item_target, item_scale,
                             item_pad_row, keys_grouped)}

    def owner_bpr_walk(self, ctx, q_all, pos_score, qgrad_all):
        state, item_local, recv_keys, n_seg, stride, pos_rows, n, mean_den, item_target, item_scale, pad, grouped = ctx['prepared']
        return self.owner_bpr_forward(state, item_local, q_all, recv_keys, n_seg, stride, pos_rows, pos_score, n, mean_den,
                                      item_target, item_scale, qgrad_all, item_pad_row=pad, keys_grouped=grouped)

    def owner_bpr_finish(self, ctx, dsum_all, parts=0):
        own = ctx['pos_rows'] >= 0
        qi = torch.nonzero(own).flatten()
        rows_p, dpos = ctx['pos_rows'][own], -dsum_all[own]
        item_local, q_all, gate = ctx['item_local'], ctx['q_all'], ctx['gate']
        if parts in (0, 1):
            ctx['qgrad_all'].index_add_(0, qi, gate * dpos.unsqueeze(1) * item_local[rows_p])    # reads the rows: before the update
        if parts in (0, 2):
            rows, qidx, d = torch.cat([ctx['rows'], rows_p]), torch.cat([ctx['qidx'], qi]), torch.cat([ctx['d'], dpos])
            live = rows != ctx['pad']
            scale = gate * (1.0 if ctx['item_scale'] is None else float(ctx['item_scale']))
            ctx['item_target'].index_add_(0, rows[live], scale * d[live].unsqueeze(1) * q_all[qidx[live]])

    def owner_ssm_forward(self, state, item_local, q_all, recv_keys, n_seg, stride, pos_rows, n, mean_den, item_target, item_scale,
                          logq_rows=None, item_pad_row=-1, keys_grouped=False):
        total = int(recv_keys.view(n_seg, stride)[:, 1].sum())
        state['step_dropped'][0] = total
        state['overflow'] += total
        self._publish_scale(state, 0.0 if total else 1.0, item_scale)
        keys = recv_keys[self._live(recv_keys, n_seg, stride)]
        rows, qidx = keys & 0xffffffff, keys >> 32
        z = (item_local[rows] * q_all[qidx]).sum(-1)
        if logq_rows is not None:
            z = z - logq_rows[rows]                                                 # loss_func.py:82
        Q = q_all.shape[0]
        run_max = torch.full((Q,), float('-inf')).scatter_reduce(0, qidx, z, 'amax', include_self=True)
        run_sum = torch.zeros(Q).index_add_(0, qidx, torch.exp(z - run_max[qidx]))
        return {'rows': rows, 'qidx': qidx, 'z': z, 'gate': 0.0 if total else 1.0, 'item_local': item_local, 'q_all': q_all,
                'pos_rows': pos_rows, 'item_target': item_target, 'item_scale': item_scale, 'pad': item_pad_row,
                'run_max': run_max, 'run_sum': run_sum, 'mean_den': mean_den}

    def owner_ssm_finish(self, ctx, lse_all, z_pos_all, qgrad_all):
        M, gate, item_local, q_all = ctx['mean_den'], ctx['gate'], ctx['item_local'], ctx['q_all']
        rows, qidx = ctx['rows'], ctx['qidx']
        d = torch.exp(ctx['z'] - lse_all[qidx]) / M                              # softmax weight / M of every negative here
        qgrad_all.index_add_(0, qidx, gate * d.unsqueeze(1) * item_local[rows])
        own = ctx['pos_rows'] >= 0
        qi = torch.nonzero(own).flatten()
        rows_p, dpos = ctx['pos_rows'][own], (torch.exp(z_pos_all[own] - lse_all[own]) - 1.0) / M
        qgrad_all.index_add_(0, qi, gate * dpos.unsqueeze(1) * item_local[rows_p])
        rows_a, q_a, d_a = torch.cat([rows, rows_p]), torch.cat([qidx, qi]), torch.cat([d, dpos])
        live = rows_a != ctx['pad']
        scale = gate * (1.0 if ctx['item_scale'] is None else float(ctx['item_scale']))
        ctx['item_target'].index_add_(0, rows_a[live], scale * d_a[live].unsqueeze(1) * q_all[q_a[live]])

    def _elements(self, pos, neg):
        return torch.cat([pos.view(-1, 1), neg], 1).reshape(-1)

    def count(self, pos, neg, plan):
        return torch.bincount(plan.owner(self._elements(pos, neg)), minlength=plan.world).to(torch.int32)

    def route(self, pos, neg, plan, query_base, starts):
        B, n = neg.shape
        ids = self._elements(pos, neg)
        owner = plan.owner(ids)
        order = torch.argsort(owner, stable=True)
        m = torch.arange(B).repeat_interleave(n + 1)
        c = torch.arange(n + 1).repeat(B)
        key = ((query_base + m) << 32) | plan.local(ids)
        position = torch.where(c == 0, m, B + m * n + (c - 1))
        return key[order], position[order]

    def score_keys(self, item_local, q_all, keys):
        live = keys >= 0                                       # negative key = empty slot: score 0
        rows, qidx = keys.clamp(min=0) & 0xffffffff, keys.clamp(min=0) >> 32
        return torch.where(live, (item_local[rows] * q_all[qidx]).sum(-1), torch.zeros(()))

    def scatter(self, scores, positions, numel):
        out = torch.zeros(numel)
        live = positions >= 0
        out[positions[live]] = scores[live]
        return out

    def gather(self, src, positions):
        return torch.where(positions >= 0, src[positions.clamp(min=0)], torch.zeros(()))

    def backward_keys(self, item_local, q_all, keys, dscore, item_grad_local, qgrad_all, item_pad_row=-1, item_scale=None):
        keep = keys >= 0
        keys, dscore = keys[keep], dscore[keep]
        rows, qidx = keys & 0xffffffff, keys >> 32
        live = rows != item_pad_row
        qgrad_all.index_add_(0, qidx, dscore.unsqueeze(1) * item_local[rows])       # reads the rows: first
        scale = 1.0 if item_scale is None else float(item_scale)
        item_grad_local.index_add_(0, rows[live], scale * dscore[live].unsqueeze(1) * q_all[qidx[live]])


    def apply_rows(self, table, ids, rows, scale, pad_row=0, gate=None):
        keep = (ids != pad_row) & (ids >= 0)                    # negative ids: dead slots of a fixed-capacity exchange
        upd = rows[keep] * float(scale)
        table.index_add_(0, ids[keep], upd if gate is None else upd * gate)

    def full_partial(self, item_local, q_all, k, want_lse, has_pad_row):
        rows = item_local[1:] if has_pad_row else item_local
        sc = q_all @ rows.t()
        lse = torch.logsumexp(sc, -1) if want_lse else None
        if not k:
            return lse, None, None
        order = torch.argsort(-sc, dim=1, stable=True)[:, :k]           # ties -> smaller row first
        return lse, torch.gather(sc, 1, order), order + 1

    def merge_lse(self, parts):
        return torch.logsumexp(parts, -1)

    def mask_history(self, score, ids, user_hist, k):
        """baseretriever.py:386-392 on sorted candidates: history items get -inf, then the k best in order."""
        hit = (ids.unsqueeze(-1) == user_hist.unsqueeze(1)).any(-1) & (ids > 0)
        score = torch.where(hit, torch.full((), float('-inf')), score)
        order = torch.argsort(-score, dim=1, stable=True)[:, :k]
        return torch.gather(score, 1, order), torch.gather(ids, 1, order)

    def merge_topk(self, vals, ids, k):
        order = torch.argsort(-vals, dim=1, stable=True)[:, :k]
        return torch.gather(vals, 1, order), torch.gather(ids, 1, order)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_items, d, B, n, result_dir, layout='block'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from recstudio_amd.shard import RowShardPlan, ShardedItemTable
        g = torch.Generator().manual_seed(5)
        item = torch.randn(n_items, d, generator=g)
        item[0] = 0
        user = torch.randn(50, d, generator=g)
        plan = RowShardPlan(n_items, world, layout=layout)
        table = ShardedItemTable(plan.take(item, rank).clone(), plan, rank, dist, backend=CheckerBackend())
        gr = torch.Generator().manual_seed(100 + rank)
        uid = torch.randint(1, 50, (B,), generator=gr)
        pos = torch.randint(1, n_items, (B,), generator=gr)
        sampler = oracle.UniformSampler(n_items)
        assert table.exchange == 'fixed' and not table._cap
        out = table.sample_and_score(user, uid, pos, n, sampler)           # exact split + calibration
        assert table._cap[(B, n, 1)] <= B * (n + 1)
        want_pos, want_neg = oracle.retriever_forward(item, user[uid], pos, out['neg_ids'])
        np.testing.assert_allclose(out['pos_score'].numpy(), want_pos.numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(out['neg_score'].numpy(), want_neg.numpy(), rtol=1e-6, atol=1e-6)
        # G-invariant negatives: the ranks' blocks, concatenated, are what ONE process draws for the whole batch from
        # the same stream (same seed, same offset) -- the negatives of a run do not depend on the number of GPUs
        blocks = [torch.zeros(B, n, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(blocks, out['neg_ids'])
        solo = CheckerBackend()
        ids1, _ = solo.sample(sampler, B * world, n, None, None, shard=(0, 1, solo.make_generator(2022, None)))
        assert torch.equal(torch.cat(blocks), ids1)
        # gradient exchange: item grads stay on the owner, query grads come home by reduce-scatter.  Like the first
        # step this one runs the FIXED-capacity exchange (equal split, self-describing segments, no counts on the host)
        out = table.sample_and_score(user, uid, pos, n, sampler, keep_route=True)
        assert 'send_counts' not in out['route'] and out['route']['slot_of'].numel() == B * (n + 1)
        want_pos, want_neg = oracle.retriever_forward(item, user[uid], pos, out['neg_ids'])
        np.testing.assert_allclose(out['pos_score'].numpy(), want_pos.numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(out['neg_score'].numpy(), want_neg.numpy(), rtol=1e-6, atol=1e-6)
        blocks2 = [torch.zeros(B, n, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(blocks2, out['neg_ids'])
        assert not torch.equal(torch.cat(blocks2), ids1)                    # the stream moved on ...
        ids2, _ = solo.sample(sampler, B * world, n, None, None, shard=(0, 1, {'seed': 2022, 'offset': table.sample_generator['offset'] - 4}))
        assert torch.equal(torch.cat(blocks2), ids2)                        # ... in lock-step with a single process
        gg = torch.Generator().manual_seed(500 + rank)
        dpos, dneg = torch.randn(B, generator=gg), torch.randn(B, n, generator=gg)
        item_grad_local = torch.zeros(plan.n_local(rank), d)
        qgrad = table.backward(out['route'], dpos, dneg, item_grad_local)
        neg_ids = out['neg_ids']
        want_q = dpos.unsqueeze(1) * item[pos] + (dneg.unsqueeze(-1) * item[neg_ids]).sum(1)
        np.testing.assert_allclose(qgrad.numpy(), want_q.numpy(), rtol=1e-5, atol=1e-5)
        # the dense item gradient summed over ALL ranks' contributions == single-process scatter-add
        contrib = torch.zeros(n_items, d)
        qv = user[uid]
        contrib.index_add_(0, pos, dpos.unsqueeze(1) * qv)
        contrib.index_add_(0, neg_ids.reshape(-1), (dneg.unsqueeze(-1) * qv.unsqueeze(1)).reshape(-1, d))
        dist.all_reduce(contrib)                        # what every rank contributed, anywhere
        contrib[0] = 0                                  # padding row: no gradient
        np.testing.assert_allclose(item_grad_local.numpy(), plan.take(contrib, rank).numpy(), rtol=1e-5, atol=1e-5)
        # skewed ids: everything owned by the last rank, and an empty segment for rank 0
        neg2 = torch.full((B, n), n_items - 1, dtype=torch.int64)
        pos2 = torch.full((B,), n_items - 2, dtype=torch.int64)
        p2, s2 = table.score_ids(user[uid], pos2, neg2)
        w2p, w2n = oracle.retriever_forward(item, user[uid], pos2, neg2)
        np.testing.assert_allclose(p2.numpy(), w2p.numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(s2.numpy(), w2n.numpy(), rtol=1e-6, atol=1e-6)
        table.check_overflow()                                                # nothing was dropped so far
        # the exact (variable-split) exchange gives the same scores
        exact = ShardedItemTable(plan.take(item, rank).clone(), plan, rank, dist, backend=CheckerBackend(), exchange='exact')
        p3, s3 = exact.score_ids(user[uid], pos2, neg2)
        assert torch.equal(p3, p2) and torch.equal(s3, s2)
        # an id distribution that outgrows a tight capacity is DETECTED: calibrate on uniform ids without slack, then
        # send everything to one owner
        tight = ShardedItemTable(plan.take(item, rank).clone(), plan, rank, dist, backend=CheckerBackend(), slack=1.0, margin=0)
        negu = torch.randint(1, n_items, (B, n), generator=gr)
        tight.score_ids(user[uid], pos, negu)
        assert B < 100 or tight._cap[(B, n, 1, 'given')] < B * (n + 1)     # (calibrated per id source: these ids were given)
        if tight._cap[(B, n, 1, 'given')] < B * (n + 1):
            w0 = tight.item_local.clone()
            p4, s4, route4 = tight.score_ids(user[uid], pos2, neg2, keep_route=True)
            # a dropped element has a defined outcome: its score reads 0, the others are exact
            kept = torch.cat([p4.view(-1, 1), s4], 1) != 0
            want = torch.cat([w2p.view(-1, 1), w2n], 1)
            assert 0 < int(kept.sum()) < kept.numel()
            np.testing.assert_allclose(torch.cat([p4.view(-1, 1), s4], 1)[kept].numpy(), want[kept].numpy(), rtol=1e-6, atol=1e-6)
            # ... and the step changes no weight anywhere: every rank knows (from the segment headers) that something
            # was dropped and scales its updates by 0 -- in-place SGD included
            assert int(tight.state['step_dropped'][0]) > 0
            qg4 = tight.backward(route4, torch.ones(B), torch.ones(B, n), tight.item_local, item_scale=torch.tensor([-0.5]))
            assert torch.equal(tight.item_local, w0) and not qg4.any()
            with pytest.raises(RuntimeError, match='did not fit'):
                tight.check_overflow()
            assert not tight._cap                                             # recalibrates on the next step
        open(os.path.join(result_dir, f'ok{rank}'), 'w').write('ok')
    finally:
        dist.destroy_process_group()


def _chunk_worker(rank, world, port, n_items, d, B, n, chunks, result_dir, layout='block'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from recstudio_amd.shard import RowShardPlan, ShardedItemTable
        g = torch.Generator().manual_seed(5)
        item = torch.randn(n_items, d, generator=g)
        item[0] = 0
        user = torch.randn(50, d, generator=g)
        plan = RowShardPlan(n_items, world, layout=layout)
        gr = torch.Generator().manual_seed(100 + rank)
        uid = torch.randint(1, 50, (B,), generator=gr)
        pos = torch.randint(1, n_items, (B,), generator=gr)
        sampler = oracle.UniformSampler(n_items)
        whole = ShardedItemTable(plan.take(item, rank).clone(), plan, rank, dist, backend=CheckerBackend())
        piped = ShardedItemTable(plan.take(item, rank).clone(), plan, rank, dist, backend=CheckerBackend(), chunks=chunks)
        gg = torch.Generator().manual_seed(500 + rank)
        dpos, dneg = torch.randn(B, generator=gg), torch.randn(B, n, generator=gg)
        for step in range(3):                                  # step 0 calibrates (both tables), 1 and 2 run fixed
            a = whole.sample_and_score(user, uid, pos, n, sampler, keep_route=True)
            b = piped.sample_and_score(user, uid, pos, n, sampler, keep_route=True)
            assert torch.equal(a['neg_ids'], b['neg_ids'])     # same job-wide stream, sliced or not
            assert torch.equal(a['pos_score'], b['pos_score']) and torch.equal(a['neg_score'], b['neg_score'])
            assert b['route']['C'] == chunks and len(b['route']['recv_keys']) == chunks
            assert (B, n, chunks) in piped._cap and (B, n, 1) not in piped._cap
            ga, gb = torch.zeros(plan.n_local(rank), d), torch.zeros(plan.n_local(rank), d)
            qa = whole.backward(a['route'], dpos, dneg, ga)
            qb = piped.backward(b['route'], dpos, dneg, gb)
            np.testing.assert_allclose(qb.numpy(), qa.numpy(), rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(gb.numpy(), ga.numpy(), rtol=1e-5, atol=1e-5)
        piped.check_overflow()
        # one batch ahead (prepare_forward / ticket), whole and sliced: the same draws, scores, loss and gradients as in place
        for ch in (1, chunks):
            plain = ShardedItemTable(plan.take(item, rank).clone(), plan, rank, dist, backend=CheckerBackend(), chunks=ch, sample_seed=31)
            ahead = ShardedItemTable(plan.take(item, rank).clone(), plan, rank, dist, backend=CheckerBackend(), chunks=ch, sample_seed=31)
            batches = [(uid.roll(k), pos.roll(k)) for k in range(3)]
            tk = ahead.prepare_forward(batches[0][1], n, sampler, fused_loss='bpr')
            for k in range(3):
                nxt = ahead.prepare_forward(batches[k + 1][1], n, sampler, fused_loss='bpr') if k < 2 else None
                u_k, p_k = batches[k]
                a = plain.sample_and_score(user, u_k, p_k, n, sampler, keep_route=True, fused_loss='bpr')
                b = ahead.sample_and_score(user, u_k, p_k, n, sampler, keep_route=True, fused_loss='bpr', ticket=tk)
                tk = nxt
                assert torch.equal(a['neg_ids'], b['neg_ids']) and torch.equal(a['loss'], b['loss'])
                assert torch.equal(a['pos_score'], b['pos_score']) and torch.equal(a['neg_score'], b['neg_score'])
                ga, gb = torch.zeros(plan.n_local(rank), d), torch.zeros(plan.n_local(rank), d)
                qa = plain.backward(a['route'], None, None, ga)
                qb = ahead.backward(b['route'], None, None, gb)
                assert torch.equal(qa, qb) and torch.equal(ga, gb)
            with pytest.raises(ValueError, match='another batch shape'):
                ahead.sample_and_score(user, uid[:B // 2], pos[:B // 2], n, sampler,
                                       ticket=dict(ahead.prepare_forward(pos, n, sampler), ready=None))
            ahead.sample_and_score(user, uid, pos, n, sampler)          # (consumes nothing of the stale ticket: a fresh route)
            ahead.check_overflow()
        # a batch the slices do not divide runs whole
        p1, s1 = piped.score_ids(user[uid[:B - 1]], pos[:B - 1], a['neg_ids'][:B - 1])
        w1p, w1n = oracle.retriever_forward(item, user[uid[:B - 1]], pos[:B - 1], a['neg_ids'][:B - 1])
        np.testing.assert_allclose(p1.numpy(), w1p.numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(s1.numpy(), w1n.numpy(), rtol=1e-6, atol=1e-6)
        # overflow inside a slice is detected like in the whole step
        tight = ShardedItemTable(plan.take(item, rank).clone(), plan, rank, dist, backend=CheckerBackend(), slack=1.0, margin=0,
                                 chunks=chunks)
        tight.score_ids(user[uid], pos, a['neg_ids'])
        Bc = B // chunks
        if tight._cap[(B, n, chunks, 'given')] < Bc * (n + 1):
            tight.score_ids(user[uid], torch.full((B,), n_items - 2), torch.full((B, n), n_items - 1))
            with pytest.raises(RuntimeError, match='did not fit'):
                tight.check_overflow()
        open(os.path.join(result_dir, f'ok{rank}'), 'w').write('ok')
    finally:
        dist.destroy_process_group()


def _full_worker(rank, world, port, n_items, d, B, k, result_dir, layout='block'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from recstudio_amd.shard import RowShardPlan, ShardedItemTable
        g = torch.Generator().manual_seed(11)
        item = torch.randn(n_items, d, generator=g)
        item[0] = 0
        plan = RowShardPlan(n_items, world, layout=layout)
        table = ShardedItemTable(plan.take(item, rank).clone(), plan, rank, dist, backend=CheckerBackend())
        q = torch.randn(B, d, generator=torch.Generator().manual_seed(200 + rank))
        lse, tv, ti = table.full_lse_topk(q, k)
        sc = q @ item[1:].t()
        np.testing.assert_allclose(lse.numpy(), torch.logsumexp(sc, -1).numpy(), rtol=1e-6, atol=1e-6)
        wv, wi = torch.topk(sc, k)
        assert torch.equal(ti, wi + 1)
        np.testing.assert_allclose(tv.numpy(), wv.numpy(), rtol=1e-6, atol=1e-6)
        lse2, _, _ = table.full_lse_topk(q, 0)
        np.testing.assert_allclose(lse2.numpy(), lse.numpy())
        _, tv2, ti2 = table.full_lse_topk(q, k, want_lse=False)
        assert torch.equal(ti2, ti)
        open(os.path.join(result_dir, f'ok{rank}'), 'w').write('ok')
    finally:
        dist.destroy_process_group()


def _train_worker(rank, world, port, n_items, d, B, n, result_dir, layout='block'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from recstudio_amd.shard import RowShardPlan, ShardedItemTable, ShardedRetriever
        g = torch.Generator().manual_seed(21)
        item = torch.randn(n_items, d, generator=g) * 0.3
        item[0] = 0
        tower_w = torch.randn(d, 8, generator=g) * 0.3
        plan = RowShardPlan(n_items, world, layout=layout)
        table = ShardedItemTable(plan.take(item, rank).clone(), plan, rank, dist, backend=CheckerBackend())
        tower = torch.nn.Linear(8, d)
        with torch.no_grad():
            tower.weight.copy_(tower_w)
            tower.bias.zero_()
        feats, poss = [], []
        for r in range(world):          # every rank can rebuild every rank's batch for the reference
            gr = torch.Generator().manual_seed(300 + r)
            feats.append(torch.randn(B, 8, generator=gr))
            poss.append(torch.randint(1, n_items, (B,), generator=gr))
        def bpr(label, pos_score, log_pos_prob, neg_score, log_neg_prob):      # loss_func.py:50-59
            return -torch.mean(torch.nn.functional.logsigmoid(pos_score.view(-1, 1) - neg_score).mean(-1))
        trainer = ShardedRetriever(table, tower, oracle.UniformSampler(n_items), bpr, n)
        loss = trainer.training_step(feats[rank], poss[rank])
        # the same step with the item rows updated in place inside the exchange (item_sgd_lr)
        table2 = ShardedItemTable(plan.take(item, rank).clone(), plan, rank, dist, backend=CheckerBackend())
        tower2 = torch.nn.Linear(8, d)
        with torch.no_grad():
            tower2.weight.copy_(tower_w)
            tower2.bias.zero_()
        trainer2 = ShardedRetriever(table2, tower2, oracle.UniformSampler(n_items), bpr, n, item_sgd_lr=0.7)
        trainer2.training_step(feats[rank], poss[rank])
        np.testing.assert_allclose(table2.item_local.numpy(), (plan.take(item, rank) - 0.7 * trainer.item_grad_local).numpy(),
                                   rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(tower2.weight.grad.numpy(), tower.weight.grad.numpy(), rtol=1e-5, atol=1e-7)
        # single-process reference on the concatenated batch with the SAME negatives
        negs = [torch.zeros(B, n, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(negs, trainer.last_neg)
        item_ref = item.clone().requires_grad_(True)
        tower_ref = torch.nn.Linear(8, d)
        with torch.no_grad():
            tower_ref.weight.copy_(tower_w)
            tower_ref.bias.zero_()
        q = tower_ref(torch.cat(feats))
        pos, neg = torch.cat(poss), torch.cat(negs)
        ps = (q * item_ref[pos]).sum(-1)
        ns = (q.unsqueeze(1) * item_ref[neg]).sum(-1)
        ref = bpr(None, ps, None, ns, None)
        ref.backward()
        total = loss.clone()
        dist.all_reduce(total)
        np.testing.assert_allclose(total.item(), ref.item(), rtol=1e-6)
        np.testing.assert_allclose(tower.weight.grad.numpy(), tower_ref.weight.grad.numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(tower.bias.grad.numpy(), tower_ref.bias.grad.numpy(), rtol=1e-4, atol=1e-6)
        want = item_ref.grad.clone()
        want[0] = 0
        np.testing.assert_allclose(trainer.item_grad_local.numpy(), plan.take(want, rank).numpy(), rtol=1e-4, atol=1e-6)
        # a replicated nn.Embedding tower: the gradient is exchanged as (ids, rows), not as a dense all-reduce
        U = 23
        emb_w = torch.randn(U, d, generator=g) * 0.3
        uids = [torch.randint(1, U, (B,), generator=torch.Generator().manual_seed(700 + r)) for r in range(world)]
        emb = torch.nn.Embedding(U, d, padding_idx=0)
        with torch.no_grad():
            emb.weight.copy_(emb_w)
        table3 = ShardedItemTable(plan.take(item, rank).clone(), plan, rank, dist, backend=CheckerBackend())
        assert not ShardedRetriever(table3, emb, oracle.UniformSampler(n_items), bpr, n).sparse_query_rows     # opt-in
        trainer3 = ShardedRetriever(table3, emb, oracle.UniformSampler(n_items), bpr, n, sparse_query_rows=True)
        trainer3.training_step(uids[rank], poss[rank])
        assert emb.weight.grad is None
        negs3 = [torch.zeros(B, n, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(negs3, trainer3.last_neg)
        w_ref = emb_w.clone().requires_grad_(True)
        q3 = w_ref[torch.cat(uids)]
        ref3 = bpr(None, (q3 * item[torch.cat(poss)]).sum(-1), None, (q3.unsqueeze(1) * item[torch.cat(negs3)]).sum(-1), None)
        ref3.backward()
        np.testing.assert_allclose(trainer3.query_grad_dense().numpy(), w_ref.grad.numpy(), rtol=1e-4, atol=1e-6)
        # applied in place: the same replica on every rank
        emb4 = torch.nn.Embedding(U, d, padding_idx=0)
        with torch.no_grad():
            emb4.weight.copy_(emb_w)
        table4 = ShardedItemTable(plan.take(item, rank).clone(), plan, rank, dist, backend=CheckerBackend())
        ShardedRetriever(table4, emb4, oracle.UniformSampler(n_items), bpr, n, query_sgd_lr=0.3).training_step(uids[rank], poss[rank])
        np.testing.assert_allclose(emb4.weight.detach().numpy(), (emb_w - 0.3 * w_ref.grad).numpy(), rtol=1e-4, atol=1e-6)
        reps = [torch.zeros(U, d) for _ in range(world)]
        dist.all_gather(reps, emb4.weight.detach().clone())
        assert all(torch.equal(reps[0], r) for r in reps)
        # the FUSED step (stock BPRLoss / SampledSoftmaxLoss: loss, mean and d loss/d score inside the exchange's home
        # kernel, no autograd over the scores) == the autograd step above, uniform and popularity sampler
        import recstudio_amd as ra
        counts = torch.arange(n_items) % 7 + 1
        for loss_cls, ref_loss, smp, on_owners in ((ra.BPRLoss, 'bpr', oracle.UniformSampler(n_items), True),
                                                   (ra.BPRLoss, 'bpr', oracle.PopularSamplerModel(counts), True),
                                                   (ra.BPRLoss, 'bpr', oracle.UniformSampler(n_items), False),
                                                   (ra.SampledSoftmaxLoss, 'ssm', oracle.PopularSamplerModel(counts), False),
                                                   (ra.SampledSoftmaxLoss, 'ssm', oracle.PopularSamplerModel(counts), True),
                                                   (ra.SampledSoftmaxLoss, 'ssm', oracle.UniformSampler(n_items), True)):
            # on_owners: the step with the loss evaluated on the owners of the negatives (no scores travel home: BPR in one
            # pass, SampledSoftmax in two phases around an 8-byte-per-query all-reduce); else the score-at-home protocol
            # (home kernel + gradient exchange)
            tbl_f = ShardedItemTable(plan.take(item, rank).clone(), plan, rank, dist, backend=CheckerBackend(), owner_loss=on_owners)
            assert tbl_f.owner_loss_ok() == on_owners
            tower_f = torch.nn.Linear(8, d)
            with torch.no_grad():
                tower_f.weight.copy_(tower_w)
                tower_f.bias.zero_()
            tr_f = ShardedRetriever(tbl_f, tower_f, smp, loss_cls(), n, keep_neg_ids=True)
            assert tr_f._fused_loss_kind() == ref_loss
            loss_f = tr_f.training_step(feats[rank], poss[rank])
            negs_f = [torch.zeros(B, n, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(negs_f, tr_f.last_neg)
            item_r = item.clone().requires_grad_(True)
            tower_r = torch.nn.Linear(8, d)
            with torch.no_grad():
                tower_r.weight.copy_(tower_w)
                tower_r.bias.zero_()
            qr = tower_r(torch.cat(feats))
            pos_r, neg_r = torch.cat(poss), torch.cat(negs_f)
            ps_r = (qr * item_r[pos_r]).sum(-1)
            ns_r = (qr.unsqueeze(1) * item_r[neg_r]).sum(-1)
            if ref_loss == 'bpr':
                ref_f = oracle.bpr_loss(ps_r, ns_r)
            else:
                lp_r = smp.compute_item_p(pos_r) if hasattr(smp, 'pop_prob') else torch.zeros(pos_r.shape)
                ln_r = smp.compute_item_p(neg_r) if hasattr(smp, 'pop_prob') else torch.zeros(neg_r.shape)
                ref_f = oracle.sampled_softmax_loss(ps_r, lp_r, ns_r, ln_r)
            ref_f.backward()
            tot_f = loss_f.clone()
            dist.all_reduce(tot_f)
            np.testing.assert_allclose(tot_f.item(), ref_f.item(), rtol=1e-5)
            np.testing.assert_allclose(tower_f.weight.grad.numpy(), tower_r.weight.grad.numpy(), rtol=1e-4, atol=1e-6)
            want_f = item_r.grad.clone()
            want_f[0] = 0
            np.testing.assert_allclose(tr_f.item_grad_local.numpy(), plan.take(want_f, rank).numpy(), rtol=1e-4, atol=1e-6)
            if on_owners:
                # ... and with plain SGD applied in place inside the owners' pass
                tbl_s = ShardedItemTable(plan.take(item, rank).clone(), plan, rank, dist, backend=CheckerBackend())
                tower_s = torch.nn.Linear(8, d)
                with torch.no_grad():
                    tower_s.weight.copy_(tower_w)
                    tower_s.bias.zero_()
                ShardedRetriever(tbl_s, tower_s, smp, loss_cls(), n, item_sgd_lr=0.4).training_step(feats[rank], poss[rank])
                np.testing.assert_allclose(tbl_s.item_local.numpy(), (plan.take(item, rank) - 0.4 * tr_f.item_grad_local).numpy(),
                                           rtol=1e-4, atol=1e-6)
                np.testing.assert_allclose(tower_s.weight.grad.numpy(), tower_f.weight.grad.numpy(), rtol=1e-5, atol=1e-7)
                # ... and three such steps one batch ahead (prepare_step / ticket): the same draws, the same weights
                runs = []
                for ahead in ((False, True) if ref_loss == 'bpr' else ()):
                    tbl_a = ShardedItemTable(plan.take(item, rank).clone(), plan, rank, dist, backend=CheckerBackend(), sample_seed=77)
                    emb_a = torch.nn.Embedding(U, d, padding_idx=0)
                    with torch.no_grad():
                        emb_a.weight.copy_(emb_w)
                    tr_a = ShardedRetriever(tbl_a, emb_a, smp, loss_cls(), n, item_sgd_lr=0.2, query_sgd_lr=0.2, keep_neg_ids=True)
                    batches = [(uids[(rank + k) % world], poss[(rank + k) % world]) for k in range(3)]
                    seen = []
                    if ahead:
                        tk = tr_a.prepare_step(*batches[0])
                        for k in range(3):
                            nxt = tr_a.prepare_step(*batches[k + 1]) if k < 2 else None
                            seen.append((tr_a.training_step(*batches[k], ticket=tk).clone(), tr_a.last_neg.clone()))
                            tk = nxt
                    else:
                        for k in range(3):
                            seen.append((tr_a.training_step(*batches[k]).clone(), tr_a.last_neg.clone()))
                    runs.append((tbl_a.item_local.clone(), emb_a.weight.detach().clone(), seen))
                if runs:
                    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
                    for (la, na), (lb, nb) in zip(runs[0][2], runs[1][2]):
                        assert torch.equal(la, lb) and torch.equal(na, nb)
                    assert not torch.equal(runs[0][0], plan.take(item, rank))
        open(os.path.join(result_dir, f'ok{rank}'), 'w').write('ok')
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,layout', [(2, 'block'), (2, 'interleaved'), (4, 'interleaved'), (8, 'block')])
def test_sharded_training_step_equals_single_process_autograd(tmp_path, world, layout):
    """ShardedRetriever.training_step on 2 / 4 / 8 ranks (item rows sharded -- contiguous blocks or interleaved rows --,
    query tower replicated + bucketed all-reduce) == autograd of the global-mean BPR loss in one process."""
    mp.spawn(_train_worker, args=(world, _free_port(), 61, 16, 7, 4, str(tmp_path), layout), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f'ok{r}') for r in range(world))


@pytest.mark.parametrize('n_items,k,layout,world', [(101, 10, 'block', 2), (7, 3, 'block', 2), (3, 1, 'block', 2),
                                                    (101, 10, 'interleaved', 2), (7, 3, 'interleaved', 2),
                                                    (3, 1, 'interleaved', 2), (101, 10, 'block', 4), (101, 10, 'interleaved', 8),
                                                    (7, 3, 'block', 8), (7, 3, 'interleaved', 8)])
def test_sharded_full_catalog_pass_equals_single_process(tmp_path, n_items, k, layout, world):
    """k larger than one shard's row count ((7, 3): shard 0 holds 3 real rows, shard 1 holds 3) and a last
    shard with a single row ((3, 1): rows_per_shard = 2 -> shard 1 = {2}); at world 8 a 7-item catalog leaves ranks with
    an empty block (block layout: rows_per_shard = 1, rank 7 holds nothing; rank 0 only the padding row)."""
    mp.spawn(_full_worker, args=(world, _free_port(), n_items, 16, 5, k, str(tmp_path), layout), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f'ok{r}') for r in range(world))


@pytest.mark.parametrize('n_items,n,B,layout,world', [(101, 5, 9, 'block', 2), (64, 1, 9, 'block', 2), (1001, 7, 500, 'block', 2),
                                                      (101, 5, 9, 'interleaved', 2), (1001, 7, 500, 'interleaved', 2),
                                                      (1001, 7, 200, 'block', 4), (1001, 7, 120, 'interleaved', 8),
                                                      (101, 5, 9, 'block', 8)])
def test_sharded_scores_equal_single_process(tmp_path, n_items, n, B, layout, world):
    """Forward + gradient exchange on 2 / 4 / 8 ranks == single process: exact-split calibration step, then the
    fixed-capacity exchange; negatives identical to a world-1 draw (G-invariance); overflow of a tight capacity detected
    (B >= 100)."""
    mp.spawn(_worker, args=(world, _free_port(), n_items, 16, B, n, str(tmp_path), layout), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f'ok{r}') for r in range(world))


@pytest.mark.parametrize('B,n,chunks,layout,world', [(8, 5, 2, 'block', 2), (12, 3, 4, 'block', 2), (400, 7, 2, 'block', 2),
                                                     (400, 7, 2, 'interleaved', 2), (120, 7, 2, 'block', 8),
                                                     (120, 7, 4, 'interleaved', 4)])
def test_pipelined_slices_equal_the_whole_step(tmp_path, B, n, chunks, layout, world):
    """ShardedItemTable(chunks=C): the step cut into C query slices with asynchronously issued exchanges gives the same
    negatives and bit-equal scores as the whole step, the same gradients (summed in a different order), keeps its
    own per-slice capacity, detects overflow, and falls back to the whole step for batches C does not divide."""
    mp.spawn(_chunk_worker, args=(world, _free_port(), 1001, 16, B, n, chunks, str(tmp_path), layout), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f'ok{r}') for r in range(world))


def test_interleaved_plan_partitions_all_rows():
    from recstudio_amd.shard import RowShardPlan
    for n_items, world in ((100_000_001, 8), (101, 2), (7, 8), (64, 4), (3, 2)):
        plan = RowShardPlan(n_items, world, layout='interleaved')
        assert sum(plan.n_local(r) for r in range(world)) == n_items and plan.rows_arg == 0
        ids = torch.tensor([0, 1, n_items // 2, n_items - 1])
        own, loc = plan.owner(ids), plan.local(ids)
        for i, o, l in zip(ids.tolist(), own.tolist(), loc.tolist()):
            assert 0 <= l < plan.n_local(o) and plan.global_ids(o, l) == i
    plan = RowShardPlan(11, 3, layout='interleaved')
    full = torch.arange(11.).view(11, 1)
    parts = [plan.take(full, r) for r in range(3)]
    assert [p.flatten().tolist() for p in parts] == [[0., 3., 6., 9.], [1., 4., 7., 10.], [2., 5., 8.]]
    assert torch.equal(plan.assemble(parts), full)
    with pytest.raises(ValueError):
        plan.bounds(0)


def test_plan_partitions_all_rows():
    from recstudio_amd.shard import RowShardPlan
    for n_items, world in ((100_000_001, 8), (101, 2), (7, 8), (64, 4)):
        plan = RowShardPlan(n_items, world)
        spans = [plan.bounds(r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n_items
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        ids = torch.tensor([0, 1, n_items // 2, n_items - 1])
        own = plan.owner(ids)
        for i, o in zip(ids.tolist(), own.tolist()):
            lo, hi = plan.bounds(o)
            assert lo <= i < hi


# ---------------------------------------------------------------------------------------------------------------------
# BaseRetriever.fit / evaluate as one rank of a multi-process job (VERDICT r2 missing #1 / #2)
def _fit_worker(rank, world, port, result_dir, epochs, batch_global, train_extra=None):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import recstudio_amd as ra
        from recstudio_amd.dataset import TripletDataset
        g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'data_ml100k.npz'))
        conf = {'train': {'epochs': epochs, 'batch_size': batch_global // world, 'negative_count': 5, 'seed': 2022,
                          'learning_rate': 0.01, 'early_stop_patience': 100},
                'eval': {'batch_size': 64 // world, 'cutoff': [10], 'val_metrics': ['ndcg', 'recall'], 'topk': 50,
                         'test_metrics': ['ndcg', 'recall']},
                'model': {'embed_dim': 16}}
        train_extra = dict(train_extra or {})
        trim = train_extra.pop('_trim', None)
        conf['train'].update(train_extra)
        model = ra.BPR(conf)                                       # seeds everything (recommender.py:34-35) ...
        ds = TripletDataset('ml-100k', {'low_rating_thres': 3.0},
                            _interactions=(g['raw_user'].astype(str), g['raw_item'].astype(str),
                                           g['raw_rating'].astype(np.float64), g['raw_time'].astype(np.float64)))
        trn, val, tst = ds.build(split_ratio=[0.8, 0.1, 0.1], shuffle=True, split_mode='user_entry')    # ... then the data
        model.sampler = oracle.UniformSampler(trn.num_items)       # the checker's stand-in for the in-kernel sampler
        if trim:
            # a sample count every world size divides: no global batch is padded (a padded batch repeats samples --
            # dataset.rank_part, torch's DistributedSampler rule -- and is not world-invariant)
            trn.data_index = trn.data_index[:len(trn.data_index) // trim * trim]
        best = model.fit(trn, val, dist=dist, shard_backend=CheckerBackend(), device='cpu')
        test = model.evaluate(tst, verbose=False)
        losses = torch.cat(model.train_losses)
        sh = model._shard
        assert tuple(model.item_encoder.weight.shape) == (sh['plan'].n_local(rank), 16)     # this rank holds its rows only
        # replicas of the tower are bit-equal; the item blocks tile the table
        tower = [torch.zeros_like(model.query_encoder.weight) for _ in range(world)]
        dist.all_gather(tower, model.query_encoder.weight.detach())
        assert all(torch.equal(tower[0], t) for t in tower)
        torch.save({'best': best, 'val': dict(model.logged_metrics), 'test': test, 'losses': losses,
                    'item': model.item_encoder.weight.detach().clone(), 'lo': sh['lo'], 'tower': tower[0],
                    'lookahead': bool(sh.get('lookahead'))},
                   os.path.join(result_dir, f'w{world}r{rank}.pt'))
    finally:
        dist.destroy_process_group()


def test_fit_two_ranks_equals_one_rank_on_ml100k(tmp_path):
    """BPR on the ml-100k fixture through ``BaseRetriever.fit`` / ``evaluate``: two ranks (rows of the item table and of its
    Adam state sharded, tower replicated, every rank on its half of each global batch, one job-wide negative stream) give
    the losses, ndcg@10 / recall@10 and weights of the one-rank run of the same global batch to 1e-5; the loss starts at
    ln 2 (xavier-initialised towers) and falls, the ranking beats chance after two epochs."""
    epochs, batch_global = 2, 2048
    for world in (1, 2):
        mp.spawn(_fit_worker, args=(world, _free_port(), str(tmp_path), epochs, batch_global), nprocs=world, join=True)
    one = torch.load(tmp_path / 'w1r0.pt', weights_only=False)
    two = [torch.load(tmp_path / f'w2r{r}.pt', weights_only=False) for r in range(2)]
    assert one['losses'].numel() == epochs * ((66868 + batch_global - 1) // batch_global)
    assert abs(float(one['losses'][0]) - 0.6931) < 2e-3 and float(one['losses'][-1]) < float(one['losses'][0]) - 0.05
    for t in two:
        np.testing.assert_allclose(t['losses'].numpy(), one['losses'].numpy(), rtol=1e-5, atol=1e-6)
        for k in ('ndcg@10', 'recall@10'):
            assert abs(t['val'][k] - one['val'][k]) < 1e-5 and abs(t['test'][k] - one['test'][k]) < 1e-5
        assert t['best'] == pytest.approx(one['best'], abs=1e-5)
        np.testing.assert_allclose(t['tower'].numpy(), one['tower'].numpy(), rtol=1e-4, atol=1e-6)
    assert one['val']['ndcg@10'] > 0.01                              # two epochs already rank better than chance
    items = torch.cat([two[0]['item'], two[1]['item']])
    assert two[0]['lo'] == 0 and two[1]['lo'] == two[0]['item'].shape[0]
    np.testing.assert_allclose(items.numpy(), one['item'].numpy(), rtol=1e-4, atol=1e-6)
    assert not items[0].any()                                        # the padding row never moves
    # the same job with train.shard_layout 'interleaved' (rank r holds rows r, r + 2, ...): the same run again
    os.makedirs(tmp_path / 'il')
    mp.spawn(_fit_worker, args=(2, _free_port(), str(tmp_path / 'il'), epochs, batch_global, {'shard_layout': 'interleaved'}),
             nprocs=2, join=True)
    il = [torch.load(tmp_path / 'il' / f'w2r{r}.pt', weights_only=False) for r in range(2)]
    for t in il:
        np.testing.assert_allclose(t['losses'].numpy(), one['losses'].numpy(), rtol=1e-5, atol=1e-6)
        for k in ('ndcg@10', 'recall@10'):
            assert abs(t['val'][k] - one['val'][k]) < 1e-5 and abs(t['test'][k] - one['test'][k]) < 1e-5
    items = torch.empty_like(one['item'])
    items[0::2], items[1::2] = il[0]['item'], il[1]['item']
    np.testing.assert_allclose(items.numpy(), one['item'].numpy(), rtol=1e-4, atol=1e-6)


def test_fit_two_ranks_clips_the_global_gradient_norm(tmp_path):
    """``train.grad_clip_norm`` under a sharded fit: the norm is taken over all shards of the item table plus the tower
    (once), so two ranks reproduce the one-rank run; the threshold is low enough to clip every step (SGD, so that the
    clipped scale shows in the weights), and the clipped run differs from the unclipped one."""
    extra = {'grad_clip_norm': 1e-3, 'learner': 'sgd', 'learning_rate': 5.0}
    for world in (1, 2):
        mp.spawn(_fit_worker, args=(world, _free_port(), str(tmp_path), 1, 2048, extra), nprocs=world, join=True)
    one = torch.load(tmp_path / 'w1r0.pt', weights_only=False)
    two = [torch.load(tmp_path / f'w2r{r}.pt', weights_only=False) for r in range(2)]
    for t in two:
        np.testing.assert_allclose(t['losses'].numpy(), one['losses'].numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(t['tower'].numpy(), one['tower'].numpy(), rtol=1e-4, atol=1e-7)
    items = torch.cat([two[0]['item'], two[1]['item']])
    np.testing.assert_allclose(items.numpy(), one['item'].numpy(), rtol=1e-4, atol=1e-7)
    os.makedirs(tmp_path / 'x')
    mp.spawn(_fit_worker, args=(1, _free_port(), str(tmp_path / 'x'), 1, 2048, dict(extra, grad_clip_norm=None)), nprocs=1, join=True)
    free = torch.load(tmp_path / 'x' / 'w1r0.pt', weights_only=False)
    assert (free['item'] - one['item']).abs().max() > 1e-4           # clipping really was in force


def test_fit_two_ranks_fused_sgd_follows_the_scheduler(tmp_path):
    """``train.fused_optimizer: 'sgd'`` (item rows updated inside the backward exchange, user rows through the row-sparse
    exchange) with ``train.scheduler: 'exponential'``: the rate the kernels get after each epoch is the scheduler's, two ranks
    reproduce the one-rank run, and the second epoch really ran at another rate."""
    extra = {'fused_optimizer': 'sgd', 'scheduler': 'exponential', 'learning_rate': 200.0}
    for world in (1, 2):
        mp.spawn(_fit_worker, args=(world, _free_port(), str(tmp_path), 2, 4096, extra), nprocs=world, join=True)
    one = torch.load(tmp_path / 'w1r0.pt', weights_only=False)
    two = [torch.load(tmp_path / f'w2r{r}.pt', weights_only=False) for r in range(2)]
    assert abs(one['val']['lr'] - 200.0 * 0.98 ** 2) < 1e-4
    for t in two:
        np.testing.assert_allclose(t['losses'].numpy(), one['losses'].numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(t['tower'].numpy(), one['tower'].numpy(), rtol=1e-4, atol=1e-6)
    items = torch.cat([two[0]['item'], two[1]['item']])
    np.testing.assert_allclose(items.numpy(), one['item'].numpy(), rtol=1e-4, atol=1e-6)
    os.makedirs(tmp_path / 'c')
    mp.spawn(_fit_worker, args=(1, _free_port(), str(tmp_path / 'c'), 2, 4096, dict(extra, scheduler=None)), nprocs=1, join=True)
    const = torch.load(tmp_path / 'c' / 'w1r0.pt', weights_only=False)
    n1 = one['losses'].numel() // 2
    assert torch.equal(const['losses'][:n1], one['losses'][:n1])                 # the first epoch: the same rate
    assert (const['item'] - one['item']).abs().max() > 1e-5                      # the second: 196 instead of 200
    # train.shard_lookahead: every step's negatives / routing / key exchange / owner sorts issued one batch ahead -- the same run
    os.makedirs(tmp_path / 'la')
    mp.spawn(_fit_worker, args=(2, _free_port(), str(tmp_path / 'la'), 2, 4096, dict(extra, shard_lookahead=True)), nprocs=2, join=True)
    la = [torch.load(tmp_path / 'la' / f'w2r{r}.pt', weights_only=False) for r in range(2)]
    for t, ref in zip(la, two):
        assert t['lookahead'] and not ref['lookahead']
        assert torch.equal(t['losses'], ref['losses']) and torch.equal(t['item'], ref['item']) and torch.equal(t['tower'], ref['tower'])



# ---------------------------------------------------------------------------------------------------------------------
# ITEM-TOWER query encoders over the sharded table (SASRec embeds its history with the table it scores
# against, seq/sasrec.py:14, :42, :107 -- the sharded fit must keep that embedding tied and must not replicate the catalog)
class _HistTower(torch.nn.Module):
    """A minimal item tower: masked mean of the history's item vectors through one dense layer."""

    def __init__(self, item_encoder, d):
        super().__init__()
        self.item_encoder = item_encoder
        self.lin = torch.nn.Linear(d, d)

    def forward(self, hist):
        rows = self.item_encoder(hist)                                     # [B, L, d]; padding rows are zero
        cnt = (hist != 0).sum(-1, keepdim=True).clamp(min=1)
        return self.lin(rows.sum(1) / cnt)


def _item_tower_worker(rank, world, port, n_items, d, B, L, n, result_dir, layout):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import recstudio_amd as ra
        from recstudio_amd.shard import RowShardPlan, ShardedItemTable, ShardedRetriever, ShardedRows
        g = torch.Generator().manual_seed(33)
        item = torch.randn(n_items, d, generator=g) * 0.3
        item[0] = 0
        lin_w, lin_b = torch.randn(d, d, generator=g) * 0.3, torch.randn(d, generator=g) * 0.1
        plan = RowShardPlan(n_items, world, layout=layout)
        hists, poss = [], []
        for r in range(world):
            gr = torch.Generator().manual_seed(900 + r)
            h = torch.randint(1, n_items, (B, L), generator=gr)
            h[torch.arange(L).view(1, -1) >= torch.randint(1, L + 1, (B, 1), generator=gr)] = 0     # right-padded, ragged
            if r == 0:
                h[0, :] = n_items - 1                                      # a whole row owned by the last rank
            hists.append(h)
            poss.append(torch.randint(1, n_items, (B,), generator=gr))

        def make(loss, **kw):
            table = ShardedItemTable(plan.take(item, rank).clone(), plan, rank, dist, backend=CheckerBackend())
            table.uniform_lookups = True                                   # every rank looks up a [B, L] window: fixed-capacity segments
            rows = ShardedRows(table)
            tower = _HistTower(rows, d)
            with torch.no_grad():
                tower.lin.weight.copy_(lin_w)
                tower.lin.bias.copy_(lin_b)
            assert [tuple(p.shape) for p in tower.parameters()] == [(d, d), (d,)]      # no [N, d] tensor in the tower
            trainer = ShardedRetriever(table, tower, oracle.UniformSampler(n_items), loss, n, keep_neg_ids=True, **kw)
            rows.bind(trainer)
            return table, tower, trainer

        # the look-up itself: rows of the full table, zero rows at the padding, any shape
        table, tower, trainer = make(ra.BPRLoss())
        got = table.lookup_rows(hists[rank])
        assert torch.equal(got, item[hists[rank]]) and not got[hists[rank] == 0].any()
        assert table.lookup_rows(torch.zeros(0, dtype=torch.int64)).shape == (0, d)
        assert not table.lookup_rows(torch.zeros(3, dtype=torch.int64)).any()          # nothing but padding: no traffic
        def bce(label, pos_score, log_pos_prob, neg_score, log_neg_prob):      # loss_func.py:100-132 under autograd
            return oracle.bce_loss(pos_score, neg_score)
        for loss, kind in ((ra.BPRLoss(), 'fused'), (bce, 'autograd')):
            table, tower, trainer = make(loss)
            assert (trainer._fused_loss_kind() is not None) == (kind == 'fused')
            loss_r = trainer.training_step(hists[rank], poss[rank], None)
            negs = [torch.zeros(B, n, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(negs, trainer.last_neg)
            # ONE process, ONE table used by the tower and by the scores (tied), the concatenated batch, same negatives
            item_ref = item.clone().requires_grad_(True)
            ref_tower = _HistTower(lambda ids: item_ref[ids], d)
            with torch.no_grad():
                ref_tower.lin.weight.copy_(lin_w)
                ref_tower.lin.bias.copy_(lin_b)
            q = ref_tower(torch.cat(hists))
            pos, neg = torch.cat(poss), torch.cat(negs)
            ps, ns = (q * item_ref[pos]).sum(-1), (q.unsqueeze(1) * item_ref[neg]).sum(-1)
            ref = oracle.bpr_loss(ps, ns) if kind == 'fused' else oracle.bce_loss(ps, ns)
            ref.backward()
            total = loss_r.clone()
            dist.all_reduce(total)
            np.testing.assert_allclose(total.item(), ref.item(), rtol=1e-5)
            np.testing.assert_allclose(tower.lin.weight.grad.numpy(), ref_tower.lin.weight.grad.numpy(), rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(tower.lin.bias.grad.numpy(), ref_tower.lin.bias.grad.numpy(), rtol=1e-4, atol=1e-6)
            want = item_ref.grad.clone()          # score-side AND tower-side gradient of the one tied table
            want[0] = 0
            np.testing.assert_allclose(trainer.item_grad_local.numpy(), plan.take(want, rank).numpy(), rtol=1e-4, atol=1e-6)
            # the tower side really contributed (else the test would pass on an untied table)
            tower_only = torch.zeros(n_items, d).index_add_(0, torch.cat(hists).reshape(-1), torch.ones(world * B * L, d))
            tower_only[0] = 0
            assert (want[tower_only[:, 0] > 0].abs().sum(-1) > 0).all()
            # in-place SGD inside the exchanges: the block moves by -lr * that gradient
            table2, tower2, trainer2 = make(loss, item_sgd_lr=0.7)
            trainer2.training_step(hists[rank], poss[rank], None)
            np.testing.assert_allclose(table2.item_local.numpy(), (plan.take(item, rank) - 0.7 * trainer.item_grad_local).numpy(),
                                       rtol=1e-4, atol=1e-6)
            trainer2.set_sgd_lr(0.1)
            assert tower2.item_encoder.grad_scale == -0.1
        # the TRAINING look-up travels in fixed-capacity segments (no host round trip in a step); one that does not fit them
        # hands zero rows to the dropped positions AND counts them into the step's dropped total: the step is gated to a no-op
        # on both sides of the tied table (score-side rows and tower-side rows), the overflow is reported, the capacity
        # recalibrated by the next step
        table3, tower3, trainer3 = make(ra.BPRLoss(), item_sgd_lr=0.7)
        trainer3.training_step(hists[rank], poss[rank], None)
        assert ('rows_frac',) in table3._cap and int(table3.state['step_dropped']) == 0
        before = table3.item_local.clone()
        table3._cap[('rows_frac',)], table3.margin = 1e-9, 0               # capacity 1 slot per owner
        trainer3.training_step(hists[rank], poss[rank], None)
        assert int(table3.state['step_dropped']) > 0 and torch.equal(table3.item_local, before)
        with pytest.raises(RuntimeError, match='did not fit'):
            table3.check_overflow()
        trainer3.training_step(hists[rank], poss[rank], None)             # recalibrated: trains again
        assert int(table3.state['step_dropped']) == 0 and table3._cap[('rows_frac',)] > 0.01 and not torch.equal(table3.item_local, before)
        open(os.path.join(result_dir, f'ok{rank}'), 'w').write('ok')
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,layout', [(2, 'block'), (2, 'interleaved'), (4, 'block')])
def test_item_tower_over_the_sharded_table_stays_tied(tmp_path, world, layout):
    """A query tower that embeds item ids itself (``shard.ShardedRows``: ids out, rows back, gradients to the owners): the
    look-up equals the full table's rows, and a training step's item gradient -- score side + tower side, summed on the
    owner -- equals autograd over ONE tied table in one process; the tower's parameters hold no [N, d] tensor."""
    mp.spawn(_item_tower_worker, args=(world, _free_port(), 67, 16, 6, 5, 3, str(tmp_path), layout), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f'ok{r}') for r in range(world))


def _sasrec_fit_worker(rank, world, port, result_dir, layout='block'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import recstudio_amd as ra
        from recstudio_amd import shard
        from recstudio_amd.dataset import SeqDataset
        g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'data_ml100k.npz'))
        conf = {'train': {'epochs': 1, 'batch_size': 4096 // world, 'negative_count': 4, 'seed': 2022, 'learning_rate': 0.01,
                          'early_stop_patience': 100, 'shard_layout': layout},
                'eval': {'batch_size': 256 // world, 'cutoff': [10], 'val_metrics': ['ndcg', 'recall'], 'topk': 50,
                         'test_metrics': ['ndcg', 'recall']},
                'model': {'embed_dim': 16, 'hidden_size': 16, 'layer_num': 1, 'head_num': 1, 'dropout_rate': 0.0}}
        # (dropout off: its masks come from each rank's own generator, so they differ with the world size)
        model = ra.SASRec(conf, loss=ra.SampledSoftmaxLoss())
        ds = SeqDataset('ml-100k', {'low_rating_thres': 3.0, 'max_seq_len': 8},
                        _interactions=(g['raw_user'].astype(str), g['raw_item'].astype(str),
                                       g['raw_rating'].astype(np.float64), g['raw_time'].astype(np.float64)))
        trn, val, tst = ds.build(split_ratio=2)
        # an even number of training samples: the last global batch then splits over two ranks without padding (a padded
        # batch repeats a sample -- dataset.rank_part, torch's DistributedSampler rule -- and is not world-invariant)
        trn.data_index = trn.data_index[:len(trn.data_index) // 2 * 2]
        model.sampler = oracle.UniformSampler(trn.num_items)
        seen = []
        orig = shard.allreduce_grads

        def spy(params, *a, **k):
            params = list(params)
            seen.extend(tuple(p.shape) for p in params)
            return orig(params, *a, **k)
        shard.allreduce_grads = spy
        best = model.fit(trn, val, dist=dist, shard_backend=CheckerBackend(), device='cpu')
        test = model.evaluate(tst, verbose=False)
        sh = model._shard
        # the tower's embedding IS the sharded table: no replica of the catalog anywhere in the tower, none all-reduced
        assert isinstance(model.query_encoder.item_encoder, shard.ShardedRows) and sh['tower_rows'] is model.query_encoder.item_encoder
        assert all(s[0] != trn.num_items for s in seen) and all(p.shape[0] != trn.num_items for p in model.query_encoder.parameters())
        assert tuple(model.item_encoder.weight.shape) == (sh['plan'].n_local(rank), 16)
        assert 'query_encoder.item_encoder.weight' not in model.state_dict()
        dense = torch.cat([p.detach().reshape(-1) for p in model.query_encoder.parameters()])
        reps = [torch.zeros_like(dense) for _ in range(world)]
        dist.all_gather(reps, dense)
        assert all(torch.equal(reps[0], t) for t in reps)                  # the dense tower replicas stay bit-equal
        torch.save({'best': best, 'val': dict(model.logged_metrics), 'test': test, 'losses': torch.cat(model.train_losses),
                    'item': model.item_encoder.weight.detach().clone(), 'tower': dense}, os.path.join(result_dir, f'w{world}r{rank}.pt'))
    finally:
        dist.destroy_process_group()


def test_sasrec_fit_two_ranks_equals_one_rank_on_ml100k(tmp_path):
    """SASRec (item tower: history embedded with the scored table) through ``BaseRetriever.fit`` / ``evaluate`` on two ranks
    == the one-rank run: losses, ndcg@10 / recall@10, the Transformer's weights and the (tied) item table; block and
    interleaved rows."""
    for world in (1, 2):
        mp.spawn(_sasrec_fit_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    os.makedirs(tmp_path / 'il')
    mp.spawn(_sasrec_fit_worker, args=(2, _free_port(), str(tmp_path / 'il'), 'interleaved'), nprocs=2, join=True)
    one = torch.load(tmp_path / 'w1r0.pt', weights_only=False)
    assert float(one['losses'][-1]) < float(one['losses'][0]) - 0.05 and one['val']['ndcg@10'] > 0.01
    for sub, il in (('', False), ('il', True)):
        two = [torch.load(tmp_path / sub / f'w2r{r}.pt', weights_only=False) for r in range(2)]
        for t in two:
            np.testing.assert_allclose(t['losses'].numpy(), one['losses'].numpy(), rtol=2e-5, atol=1e-6)
            for k in ('ndcg@10', 'recall@10'):
                assert abs(t['val'][k] - one['val'][k]) < 1e-4 and abs(t['test'][k] - one['test'][k]) < 1e-4
            np.testing.assert_allclose(t['tower'].numpy(), one['tower'].numpy(), rtol=1e-3, atol=5e-5)
        items = torch.empty_like(one['item'])
        if il:
            items[0::2], items[1::2] = two[0]['item'], two[1]['item']
        else:
            items = torch.cat([two[0]['item'], two[1]['item']])
        np.testing.assert_allclose(items.numpy(), one['item'].numpy(), rtol=1e-3, atol=1e-5)
        assert not items[0].any()


def test_fit_four_and_eight_ranks_equal_one_rank_on_ml100k(tmp_path):
    """The target world size: ``BaseRetriever.fit`` / ``evaluate`` as 4 and as 8 real processes -- bank /
    cursor / header logic, rank parts of every batch, the sharded top-k with history mask, metric all-reduce -- reproduce
    the one-rank run of the same global batches (BPR, ml-100k fixture, one epoch; interleaved rows at world 8)."""
    epochs, batch_global = 1, 2048
    runs = ((1, {'_trim': 8}), (4, {'_trim': 8}), (8, {'_trim': 8, 'shard_layout': 'interleaved'}))
    for world, extra in runs:
        mp.spawn(_fit_worker, args=(world, _free_port(), str(tmp_path), epochs, batch_global, extra), nprocs=world, join=True)
    one = torch.load(tmp_path / 'w1r0.pt', weights_only=False)
    for world, extra in runs[1:]:
        parts = [torch.load(tmp_path / f'w{world}r{r}.pt', weights_only=False) for r in range(world)]
        for t in parts:
            np.testing.assert_allclose(t['losses'].numpy(), one['losses'].numpy(), rtol=1e-5, atol=1e-6)
            for k in ('ndcg@10', 'recall@10'):
                assert abs(t['val'][k] - one['val'][k]) < 1e-5 and abs(t['test'][k] - one['test'][k]) < 1e-5
            np.testing.assert_allclose(t['tower'].numpy(), one['tower'].numpy(), rtol=1e-4, atol=1e-6)
        if 'shard_layout' in extra:
            items = torch.empty_like(one['item'])
            for r, t in enumerate(parts):
                items[r::world] = t['item']
        else:
            items = torch.cat([t['item'] for t in parts])
            assert [t['lo'] for t in parts] == [r * parts[0]['item'].shape[0] for r in range(world)]
        np.testing.assert_allclose(items.numpy(), one['item'].numpy(), rtol=1e-4, atol=1e-5)     # (Adam: a different summation order on a near-zero gradient)
        assert not items[0].any()


def _wide_topk_worker(rank, world, port, result_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import recstudio_amd as ra
        from recstudio_amd import ops
        from recstudio_amd.shard import RowShardPlan, ShardedItemTable
        ops.FULLSCORE_MAX_K = 8                       # (the in-kernel select limit, shrunk so that a 41-item catalog crosses it)
        n_items, d, B, k, width = 41, 8, 5, 3, 10
        g = torch.Generator().manual_seed(3)
        item = torch.randn(n_items, d, generator=g)
        item[0] = 0
        plan = RowShardPlan(n_items, world)
        table = ShardedItemTable(plan.take(item, rank).clone(), plan, rank, dist, backend=CheckerBackend())
        model = ra.BaseRetriever({'train': {'seed': None}})
        model._shard = {'table': table, 'n_items': n_items, 'hist_width': width, 'topk_narrow': 6}
        q = torch.randn(B, d, generator=torch.Generator().manual_seed(50 + rank))
        calls = []
        full = table.full_lse_topk
        table.full_lse_topk = lambda qq, kk, want_lse=True: (calls.append(kk), full(qq, kk, want_lse))[1]
        best = torch.argsort(-(q @ item[1:].t()), dim=1) + 1
        for blocked in (False, True):
            hist = torch.zeros(B, width, dtype=torch.int64)
            hist[:, :4] = best[:, 10:14]                                   # histories that do not touch the head
            if blocked and rank == world - 1:
                hist[2, :6] = best[2, :6]      # ONE row of ONE rank whose history is the whole narrow candidate list
            del calls[:]
            score, ids = model._topk_sharded(q, k, hist, False)
            # k + |history| = 13 > 8: six candidates first; every rank goes wide exactly when some row somewhere ran short
            assert calls == ([6, 13] if blocked else [6])
            want_s, want_i = oracle.topk_with_history(q, item, k, hist)
            assert torch.equal(ids, want_i)
            np.testing.assert_allclose(score.numpy(), want_s.numpy(), rtol=1e-6, atol=1e-6)
        open(os.path.join(result_dir, f'ok{rank}'), 'w').write('ok')
    finally:
        dist.destroy_process_group()


def test_sharded_topk_with_histories_longer_than_the_select(tmp_path):
    """``k + |history|`` beyond the in-kernel select (ml-1m: ~1.8 k) used to raise after a full epoch.  Now the
    single-process strategy with rank-uniform shapes: a narrow candidate list, history dropped, and the wide pass only
    when a row of ANY rank is left short -- decided by one all-reduced flag, so every rank takes the same branch."""
    world = 2
    mp.spawn(_wide_topk_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f'ok{r}') for r in range(world))


def test_fit_with_device_initialised_blocks_is_world_invariant(tmp_path):
    """``train.shard_init: 'device'`` (the default once the table exceeds 2 GiB): no rank ever holds the full item table -- its
    rows are drawn on the owner's device from per-chunk generator streams -- and the run is still the one-rank run."""
    extra = {'shard_init': 'device', '_trim': 2}
    for world in (1, 2):
        mp.spawn(_fit_worker, args=(world, _free_port(), str(tmp_path), 1, 2048, extra), nprocs=world, join=True)
    one = torch.load(tmp_path / 'w1r0.pt', weights_only=False)
    two = [torch.load(tmp_path / f'w2r{r}.pt', weights_only=False) for r in range(2)]
    for t in two:
        np.testing.assert_allclose(t['losses'].numpy(), one['losses'].numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(t['tower'].numpy(), one['tower'].numpy(), rtol=1e-4, atol=1e-6)
    items = torch.cat([two[0]['item'], two[1]['item']])
    np.testing.assert_allclose(items.numpy(), one['item'].numpy(), rtol=1e-4, atol=1e-5)
    assert not items[0].any() and float(items.std()) > 0
    # ... and it is NOT the host-initialised table (another stream): the mode really was in force
    os.makedirs(tmp_path / 'h')
    mp.spawn(_fit_worker, args=(1, _free_port(), str(tmp_path / 'h'), 1, 2048, {'shard_init': 'host', '_trim': 2}), nprocs=1, join=True)
    host = torch.load(tmp_path / 'h' / 'w1r0.pt', weights_only=False)
    assert (host['item'] - one['item']).abs().max() > 1e-3
