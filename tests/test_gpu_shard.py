"""GPU: the HIP routing kernels of the sharded path against the checker backend used by the gloo
test, and a world_size-1 RCCL run of the whole sharded step against the unsharded fused forward."""
import os
import sys

import numpy as np
import pytest
import torch

import oracle
from test_shard_gloo import CheckerBackend, _free_port

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('layout', ['block', 'interleaved'])
@pytest.mark.parametrize('world,B,n', [(4, 33, 7), (8, 1000, 64), (2, 5, 1), (3, 17, 100)])
def test_route_kernels_match_checker(world, B, n, layout):
    from recstudio_amd.shard import HipBackend, RowShardPlan
    n_items = 10007
    plan = RowShardPlan(n_items, world, layout=layout)
    g = torch.Generator().manual_seed(B + n)
    pos = torch.randint(0, n_items, (B,), generator=g)
    neg = torch.randint(0, n_items, (B, n), generator=g)
    hb, cb = HipBackend(), CheckerBackend()
    counts = hb.count(pos.to(DEV), neg.to(DEV), plan).cpu()
    want_counts = cb.count(pos, neg, plan)
    assert torch.equal(counts, want_counts)
    starts = torch.tensor([0] + counts.tolist()[:-1]).cumsum(0)
    keys, positions = hb.route(pos.to(DEV), neg.to(DEV), plan, 1000, starts)
    wkeys, wpos = cb.route(pos, neg, plan, 1000, starts)
    keys, positions = keys.cpu(), positions.cpu()
    # same multiset of (key, position) pairs inside every owner segment (order inside is free)
    ends = counts.cumsum(0).tolist()
    for s, e in zip([0] + ends[:-1], ends):
        got = sorted(zip(keys[s:e].tolist(), positions[s:e].tolist()))
        want = sorted(zip(wkeys[s:e].tolist(), wpos[s:e].tolist()))
        assert got == want
    assert sorted(positions.tolist()) == list(range(B * (n + 1)))
    src = torch.randn(B * (n + 1))
    out = hb.scatter(src.to(DEV), positions.to(DEV), B * (n + 1)).cpu()
    assert torch.equal(out[positions], src)


@pytest.mark.parametrize('world,B,n,chunks,cap_frac,banks', [(4, 33, 7, 1, 2.0, 1), (8, 1000, 64, 1, 1.1, 1), (2, 6, 1, 2, 1.0, 1),
                                                             (3, 16, 100, 4, 0.5, 1), (1, 64, 64, 1, 0.7, 1),
                                                             (8, 1000, 64, 1, 1.3, 8), (2, 700, 128, 2, 0.6, 4), (1, 300, 64, 1, 4.0, 8),
                                                             (2, 41, 200, 1, 0.8, 2), (3, 9, 600, 1, 1.5, 1)])
@pytest.mark.parametrize('layout', ['block', 'interleaved'])
def test_sample_route_kernel_matches_checker(world, B, n, chunks, cap_frac, banks, layout):
    """rsa_shard_sample_route on given ids: every (slice, owner) segment holds the same multiset of keys as the
    checker's (order inside a segment is free), its header says how many are live and how many elements the rank
    dropped in the whole step; slot_of points every kept element at its own key and is -1 for a dropped one; the exact
    counts of the count-only mode are the checker's.  Then the consumers of such segments: the segment form of the
    scoring kernel, the unpack kernel, the home kernel (scores, BPR / SampledSoftmax with dropped elements left out,
    routed-order gradient) against the checker."""
    from recstudio_amd import _native as nat
    from recstudio_amd.shard import HipBackend, RowShardPlan
    from recstudio_amd._native import ptr
    import recstudio_amd as ra
    n_items, rank = 10007, 1 if world > 1 else 0
    plan = RowShardPlan(n_items, world, layout=layout)
    g = torch.Generator().manual_seed(B + n)
    pos = torch.randint(0, n_items, (B,), generator=g)
    neg = torch.randint(0, n_items, (B, n), generator=g)
    hb, cb = HipBackend(), CheckerBackend()
    st, wst = hb.new_state(DEV), cb.new_state(None)
    counts = hb.sample_route(st, plan, rank, pos.to(DEV), n, chunks, 0, None, None, neg=neg.to(DEV), count_only=True,
                             banks=banks).cpu()
    wcounts = cb.sample_route(wst, plan, rank, pos, n, chunks, 0, None, None, neg=neg, count_only=True)
    # a slice (and a bank) of the kernel is a range of its workgroups, the checker's slice a range of queries: per
    # owner the totals agree
    assert torch.equal(counts.view(chunks, world, banks).sum((0, 2)), wcounts.view(chunks, world).sum(0))
    cap = max(1, int((B // chunks) * (n + 1) / (world * banks) * cap_frac))
    r = hb.sample_route(st, plan, rank, pos.to(DEV), n, chunks, cap, None, None, neg=neg.to(DEV), banks=banks)
    stride, segs = r['stride'], chunks * world * banks
    assert stride == cap + hb.HDR
    send, slot_of = r['send'].cpu().view(segs, stride), r['slot_of'].cpu().long()
    dropped = int((counts.long() - cap).clamp(min=0).sum())
    assert torch.equal(send[:, 0], counts.long().clamp(max=cap))
    word1 = send[:, 1].view(chunks * world, banks)              # the dropped total rides in bank 0 of every (slice, owner)
    assert (word1[:, 0] == dropped).all() and not word1[:, 1:].any()
    kept = slot_of >= 0
    assert int((~kept).sum()) == dropped
    assert slot_of[kept].unique().numel() == int(kept.sum())
    # every kept element points at its own key, inside a segment of its owner, below that segment's live count
    ids_flat = torch.cat([pos.view(-1, 1), neg], 1).reshape(-1)
    m = torch.arange(B).repeat_interleave(n + 1)
    want_key = ((rank * B + m) << 32) | plan.local(ids_flat)
    assert torch.equal(r['send'].cpu()[slot_of[kept]], want_key[kept])
    seg_of, within = slot_of[kept] // stride, slot_of[kept] % stride
    assert torch.equal((seg_of // banks) % world, plan.owner(ids_flat)[kept])
    assert (within >= hb.HDR).all() and (within - hb.HDR < send[seg_of, 0]).all()
    assert torch.equal(torch.bincount(seg_of, minlength=segs), send[:, 0])
    if chunks == 1 and banks == 1 and not dropped:     # exactly the checker's keys, segment by segment
        w = cb.sample_route(wst, plan, rank, pos, n, 1, cap, None, None, neg=neg)
        wsend = w['send'].view(segs, stride)
        assert torch.equal(send[:, :2], wsend[:, :2])
        for sgm in range(segs):
            live = int(send[sgm, 0])
            assert sorted(send[sgm, hb.HDR:hb.HDR + live].tolist()) == sorted(wsend[sgm, hb.HDR:hb.HDR + live].tolist())
    # the cursors reset themselves: a second launch gives the same headers
    r2 = hb.sample_route(st, plan, rank, pos.to(DEV), n, chunks, cap, None, None, neg=neg.to(DEV), banks=banks)
    assert torch.equal(r2['send'].cpu().view(segs, stride)[:, :2], send[:, :2])
    # ---- consumers.  Owner side: the received buffer of one slice = its `world` segments
    item = torch.randn(plan.rows_per_shard, 64, device=DEV)
    q_all = torch.randn(rank * B + B, 64, device=DEV)
    per, nseg = world * banks * stride, world * banks
    fl, wfl = hb.new_state(DEV), cb.new_state(None)
    kd = r['send'][:per].contiguous()
    sc = hb.score_segments(fl, item, q_all, kd, nseg, stride).cpu()
    want = cb.score_segments(wfl, item.cpu(), q_all.cpu(), kd.cpu(), nseg, stride)
    livem = cb._live(kd.cpu(), nseg, stride)
    np.testing.assert_allclose(sc[livem], want[livem], rtol=1e-4, atol=1e-5)
    assert int(fl['step_dropped']) == int(wfl['step_dropped']) == world * dropped == int(fl['overflow'])
    rows, qidx = torch.empty_like(kd), torch.empty_like(kd)
    scale = torch.full((2,), 7.0, device=DEV)
    lr = torch.tensor([-0.25], device=DEV)
    nat.check(nat.lib().rsa_shard_unpack_segments(ptr(kd), nseg, stride, ptr(rows), ptr(qidx), ptr(lr), ptr(fl['step_dropped']),
                                                  ptr(scale), ra.ops._stream()), 'unpack')
    assert ((rows.cpu() < 0) == ~livem).all() and ((qidx.cpu() < 0) == ~livem).all()
    assert torch.equal(rows.cpu()[livem], kd.cpu()[livem] & 0xffffffff) and torch.equal(qidx.cpu()[livem], kd.cpu()[livem] >> 32)
    assert scale.tolist() == ([0.0, 0.0] if dropped else [-0.25, 1.0])
    # Home side: stand-in scores in the geometry of the send buffer
    sc_home = torch.randn(segs * stride, generator=g)
    lpp, lnp = torch.randn(B, generator=g), torch.randn(B, n, generator=g)
    o = hb.home(sc_home.to(DEV), r['slot_of'], B, n)
    so = slot_of.view(B, n + 1)
    wsc = torch.where(so >= 0, sc_home[so.clamp(min=0)], torch.zeros(()))
    assert torch.equal(o['pos_score'].cpu(), wsc[:, 0]) and torch.equal(o['neg_score'].cpu(), wsc[:, 1:])
    for loss in ('bpr', 'ssm'):
        kw = dict(loss=loss, pos_logp=lpp if loss == 'ssm' else None, neg_logp=lnp if loss == 'ssm' else None, mean_den=3 * B,
                  want_grad=True, want_dsend=True)
        got = hb.home(sc_home.to(DEV), r['slot_of'], B, n, **{k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in kw.items()})
        ref = cb.home(sc_home, r['slot_of'].cpu(), B, n, **kw)
        np.testing.assert_allclose(got['loss'].cpu(), ref['loss'], rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(got['row_loss'].cpu(), ref['row_loss'], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(got['dpos'].cpu(), ref['dpos'], rtol=1e-4, atol=1e-8)
        np.testing.assert_allclose(got['dneg'].cpu(), ref['dneg'], rtol=1e-4, atol=1e-8)
        ds, wds = got['d_send'].cpu(), ref['d_send']
        assert torch.equal(torch.isnan(wds), ~torch.zeros_like(wds, dtype=torch.bool).index_fill_(0, slot_of[kept], True))
        np.testing.assert_allclose(ds[slot_of[kept]], wds[slot_of[kept]], rtol=1e-4, atol=1e-8)
        # a dropped element sends nothing and has zero gradient; a query whose positive was dropped leaves the step
        dflat = torch.cat([got['dpos'].cpu().view(-1, 1), got['dneg'].cpu()], 1).reshape(-1)
        assert not dflat[~kept].any()
        gone = so[:, 0] < 0
        assert not got['row_loss'].cpu()[gone].any() and not got['dneg'].cpu()[gone].any()


def test_interleaved_rows_have_no_hot_shard_under_popularity_ordered_ids():
    """A Zipf catalog whose ids are in popularity order (id 1 the hottest): with contiguous blocks the owner of the head
    receives several times the mean -- and the fixed capacity is the maximum over the owners --, with interleaved rows
    every owner receives the mean +- 3 % (sampling noise of 526 k elements).  Same draw (the in-kernel popularity sampler), same counting kernel."""
    import recstudio_amd as ra
    from recstudio_amd.shard import HipBackend, RowShardPlan
    N, world, B, n = 1_000_001, 8, 2048, 256
    counts = torch.zeros(N)
    counts[1:] = (1e9 / (torch.arange(1, N, dtype=torch.float64) + 1000.0)).float()     # shifted Zipf: no single item dominates
    ps = ra.PopularSamplerModel(counts, mode=2).to(DEV)         # sampler.py:233-234: count ** 0.75 (mode 0 takes the log: nearly flat)
    hb = HipBackend()
    spec = hb.sampler_spec(ps)
    pos = torch.multinomial(counts, B, replacement=True).to(DEV)
    ratio = {}
    for layout in ('block', 'interleaved'):
        plan = RowShardPlan(N, world, layout=layout)
        gen = torch.Generator(device=DEV).manual_seed(9)
        c = hb.sample_route(hb.new_state(DEV), plan, 0, pos, n, 1, 0, spec, gen, count_only=True).cpu().double()
        assert int(c.sum()) == B * (n + 1)
        ratio[layout] = float(c.max() / c.mean())
    assert ratio['block'] > 2.5 and ratio['interleaved'] < 1.03, ratio


def test_sorted_scatter_drops_negative_ids():
    """rsa_rows_update_sorted with ids < 0 (empty slots of the fixed-capacity exchange): nothing read or written for
    them -- the rest equals the scatter of the live elements alone, also when the empty slots are a long run."""
    import recstudio_amd as ra
    torch.manual_seed(0)
    N, d, m = 3001, 128, 5000
    q = torch.randn(700, d, device=DEV)
    ids = torch.randint(0, N, (m, 1), device=DEV)
    qidx = torch.randint(0, 700, (m,), device=DEV)
    coef = torch.randn(m, 1, device=DEV)
    dead = torch.rand(m, device=DEV) < 0.4
    ids_d, qidx_d = ids.clone(), qidx.clone()
    ids_d[dead] = -1
    qidx_d[dead] = -1
    got = ra.ops.scatter_rows_sorted(torch.zeros(N, d, device=DEV), q, ids_d, coef, query_index=qidx_d, pad_row=-1)
    live = ~dead
    want = ra.ops.scatter_rows_sorted(torch.zeros(N, d, device=DEV), q, ids[live].contiguous(), coef[live].contiguous(),
                                      query_index=qidx[live].contiguous(), pad_row=-1)
    assert torch.equal(got, want)
    ref = torch.zeros(N, d, device=DEV).index_add_(0, ids[live].view(-1), coef[live] * q[qidx[live]])
    np.testing.assert_allclose(got.cpu(), ref.cpu(), rtol=1e-4, atol=1e-5)


from staged_dist import StagedDist       # noqa: E402  (tools/staged_dist.py)


def _check_pipelined_equals_whole(ra, ShardedItemTable, make_table, user, uid, pos, n, sampler, rows, d, dev):
    """ShardedItemTable(chunks=C) against the whole step on the same job-wide stream: same negatives, bit-equal scores,
    same gradients up to summation order; C slices in the route; per-slice capacity."""
    B = uid.numel()
    for chunks in (2, 4):
        whole, piped = make_table(chunks=1), make_table(chunks=chunks)
        for step in range(3):
            a = whole.sample_and_score(user, uid, pos, n, sampler, keep_route=True)
            b = piped.sample_and_score(user, uid, pos, n, sampler, keep_route=True)
            assert torch.equal(a['neg_ids'], b['neg_ids'])
            assert torch.equal(a['pos_score'], b['pos_score']) and torch.equal(a['neg_score'], b['neg_score'])
            assert b['route']['C'] == chunks and len(b['route']['recv_keys']) == chunks and (B, n, chunks) in piped._cap
            _, dpos, dneg, _ = ra.ops.pairwise_loss(ra._native.LOSS_BPR, a['pos_score'], a['neg_score'])
            ga, gb = torch.zeros(rows, d, device=dev), torch.zeros(rows, d, device=dev)
            qa = whole.backward(a['route'], dpos, dneg, ga)
            qb = piped.backward(b['route'], dpos, dneg, gb)
            np.testing.assert_allclose(qb.cpu(), qa.cpu(), rtol=2e-4, atol=1e-8)
            np.testing.assert_allclose(gb.cpu(), ga.cpu(), rtol=2e-4, atol=1e-8)
        piped.check_overflow()


def _two_rank_worker(rank, world, port, backend, result_dir, layout='block'):
    import torch.distributed as dist
    import recstudio_amd as ra
    from recstudio_amd.shard import RowShardPlan, ShardedItemTable, ShardedRetriever
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dev = torch.device('cuda', rank if backend == 'nccl' else 0)
    torch.cuda.set_device(dev)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        comm = dist
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
        comm = StagedDist(dist)
    def gather_cpu(t):          # every rank's `t` (NCCL moves device tensors only)
        parts = [torch.empty_like(t if backend == 'nccl' else t.cpu()) for _ in range(world)]
        dist.all_gather(parts, t if backend == 'nccl' else t.cpu())
        return [p.cpu() for p in parts]

    def sum_cpu(t):
        x = t.clone() if backend == 'nccl' else t.cpu().clone()
        dist.all_reduce(x)
        return x.cpu()
    try:
        N, U, d, B, n = 30_011, 500, 128, 257, 64
        g = torch.Generator().manual_seed(3)
        item = (torch.randn(N, d, generator=g) * 0.1)
        item[0] = 0
        user = (torch.randn(U, d, generator=g) * 0.1).to(dev)
        counts = (torch.rand(N, generator=g) ** 3 * 100).long()
        gr = torch.Generator().manual_seed(50 + rank)
        uid = torch.randint(1, U, (B,), generator=gr).to(dev)
        pos = torch.randint(1, N, (B,), generator=gr).to(dev)
        plan = RowShardPlan(N, world, layout=layout)
        item_d = item.to(dev)
        for si, sampler in enumerate((ra.UniformSampler(N), ra.PopularSamplerModel(counts).to(dev))):
            table = ShardedItemTable(plan.take(item_d, rank).contiguous(), plan, rank, comm, sample_seed=17 + si)
            for step in range(3):                       # step 0: exact split + calibration; 1, 2: fixed capacity
                out = table.sample_and_score(user, uid, pos, n, sampler, keep_route=True)
                assert 'send_counts' not in out['route'] and (B, n, 1) in table._cap      # fixed exchange from the first step on
                ids = out['neg_ids']
                want_p, want_n = oracle.retriever_forward(item, user.cpu()[uid.cpu()], pos.cpu(), ids.cpu())
                np.testing.assert_allclose(out['neg_score'].cpu(), want_n, rtol=1e-4, atol=1e-6)
                np.testing.assert_allclose(out['pos_score'].cpu(), want_p, rtol=1e-4, atol=1e-6)
                # G-invariance on the device stream: the two ranks' blocks == one draw of [2B, n] from the same state
                blocks = gather_cpu(ids)
                gen1 = torch.Generator(device=dev).manual_seed(17 + si)
                for _ in range(step + 1):
                    with ra.rng.sharded_stream(0, 1, gen1):
                        one = sampler(torch.empty(world * B, 1, device=dev), n, None)[0]
                assert torch.equal(torch.cat(blocks), one.cpu())
                # gradient exchange == the unsharded backward kernels on the full table
                loss, dpos, dneg, _ = ra.ops.pairwise_loss(ra._native.LOSS_BPR, out['pos_score'], out['neg_score'])
                ig = torch.zeros(plan.n_local(rank), d, device=dev)
                qg = table.backward(out['route'], dpos, dneg, ig)
                ig2, _, qg2 = ra.ops.fused_backward(item_d, user, ids, dneg, query_index=uid, pos_ids=pos, dpos=dpos)
                np.testing.assert_allclose(qg.cpu(), qg2.cpu(), rtol=2e-4, atol=1e-8)
                total = sum_cpu(ig2)                     # this rank's contribution to every row, summed over the ranks
                np.testing.assert_allclose(ig.cpu(), plan.take(total, rank), rtol=2e-4, atol=1e-8)
            table.check_overflow()
        # the step cut into query slices with asynchronously issued exchanges
        _check_pipelined_equals_whole(ra, ShardedItemTable,
                                      lambda chunks: ShardedItemTable(plan.take(item_d, rank).contiguous(), plan, rank, comm,
                                                                      sample_seed=5, chunks=chunks),
                                      user, uid[:256].contiguous(), pos[:256].contiguous(), n, ra.UniformSampler(N),
                                      plan.n_local(rank), d, dev)
        # the sharded full-catalog pass
        q = torch.randn(70, d, generator=gr).to(dev) * 0.2
        table = ShardedItemTable(plan.take(item_d, rank).contiguous(), plan, rank, comm)
        lse, tv, ti = table.full_lse_topk(q, 20)
        _, wl, wv, wi = ra.ops.fullscore(item_d, q, want_lse=True, k=20)
        assert torch.equal(ti, wi)
        np.testing.assert_allclose(tv.cpu(), wv.cpu(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(lse.cpu(), wl.cpu(), rtol=1e-6)
        # a training step: global-mean BPR loss, replicated tower summed over ranks, in-place item SGD
        tower = torch.nn.Embedding(U, d).to(dev)
        with torch.no_grad():
            tower.weight.copy_(user)
        tbl = ShardedItemTable(plan.take(item_d, rank).contiguous().clone(), plan, rank, comm)
        trainer = ShardedRetriever(tbl, tower, ra.UniformSampler(N), ra.BPRLoss(), 64, item_sgd_lr=0.5, sparse_query_rows=True)
        assert trainer._fused_loss_kind() == 'bpr'           # loss + routed-order gradient inside the home kernel
        l0 = trainer.training_step(uid, pos)
        tot = sum_cpu(l0.detach().reshape(1))
        assert abs(float(tot) - 0.6931) < 0.05 and tower.weight.grad is None       # nn.Embedding tower: row-sparse gradient
        ids_all, rows_all = trainer.query_rows
        assert ids_all.shape == (world * B,) and rows_all.shape == (world * B, d) and torch.isfinite(rows_all).all()
        assert torch.equal(ids_all[rank * B:(rank + 1) * B], uid)
        # every rank holds the same rows (what it applies keeps the replicas bit-equal)
        both = gather_cpu(rows_all)
        assert torch.equal(both[0], both[1])
        # the same step with the user rows updated in place: W - lr * dense(row-sparse gradient), identical on both ranks
        tower2 = torch.nn.Embedding(U, d).to(dev)
        with torch.no_grad():
            tower2.weight.copy_(user)
        tbl2 = ShardedItemTable(plan.take(item_d, rank).contiguous().clone(), plan, rank, comm)
        trainer2 = ShardedRetriever(tbl2, tower2, ra.UniformSampler(N), ra.BPRLoss(), 64, item_sgd_lr=0.5, query_sgd_lr=0.25)
        trainer2.training_step(uid, pos)
        np.testing.assert_allclose(tower2.weight.detach().cpu(), (user - 0.25 * trainer.query_grad_dense()).cpu(),
                                   rtol=1e-5, atol=1e-7)
        reps = gather_cpu(tower2.weight.detach())
        assert torch.equal(reps[0], reps[1])
        # deterministic router at world size 2: two runs of an in-place SGD step agree bit for bit (rows, user rows, loss)
        det = []
        for _ in range(2):
            tower_d = torch.nn.Embedding(U, d).to(dev)
            with torch.no_grad():
                tower_d.weight.copy_(user)
            tbl_d = ShardedItemTable(plan.take(item_d, rank).contiguous().clone(), plan, rank, comm, sample_seed=37, deterministic=True)
            tr_d = ShardedRetriever(tbl_d, tower_d, ra.UniformSampler(N), ra.BPRLoss(), 64, item_sgd_lr=0.5, query_sgd_lr=0.25)
            ld = tr_d.training_step(uid, pos)
            tbl_d.check_overflow()
            det.append((ld.clone(), tbl_d.item_local.clone(), tower_d.weight.detach().clone()))
        assert all(torch.equal(x, y) for x, y in zip(det[0], det[1]))
        # SampledSoftmax on the owners (two phases around the (max, sum) all-reduce) == the score-at-home protocol of the same
        # step at world size 2: same draws, same loss, same item rows and query rows after one in-place SGD step
        outs = []
        for own in (True, False):
            tower_s = torch.nn.Embedding(U, d).to(dev)
            with torch.no_grad():
                tower_s.weight.copy_(user)
            tbl_s = ShardedItemTable(plan.take(item_d, rank).contiguous().clone(), plan, rank, comm, sample_seed=31)
            tr_s = ShardedRetriever(tbl_s, tower_s, ra.PopularSamplerModel((torch.arange(N) % 11 + 1)).to(dev), ra.SampledSoftmaxLoss(),
                                    64, item_sgd_lr=0.5, query_sgd_lr=0.25, keep_neg_ids=True, owner_ssm=own)
            assert tbl_s.ssm_owner_ok()
            ls = tr_s.training_step(uid, pos)
            tbl_s.check_overflow()
            outs.append((float(sum_cpu(ls.detach().reshape(1))), tr_s.last_neg.clone(), tbl_s.item_local.clone(),
                         tower_s.weight.detach().clone()))
        assert torch.equal(outs[0][1], outs[1][1])
        np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-5)
        np.testing.assert_allclose(outs[0][2].cpu(), outs[1][2].cpu(), rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(outs[0][3].cpu(), outs[1][3].cpu(), rtol=1e-4, atol=1e-7)
        assert (outs[0][2] - plan.take(item_d, rank)).abs().max() > 1e-5
        # three steps one batch ahead (prepare_step on the second stream / ticket) == the same three steps in place
        runs = []
        batches = [(uid.roll(k), pos.roll(2 * k)) for k in range(3)]
        for ahead in (False, True):
            tower3 = torch.nn.Embedding(U, d).to(dev)
            with torch.no_grad():
                tower3.weight.copy_(user)
            tbl3 = ShardedItemTable(plan.take(item_d, rank).contiguous().clone(), plan, rank, comm, sample_seed=23)
            tr3 = ShardedRetriever(tbl3, tower3, ra.UniformSampler(N), ra.BPRLoss(), 64, item_sgd_lr=0.5, query_sgd_lr=0.25,
                                   keep_neg_ids=True)
            seen = []
            if ahead:
                tk = tr3.prepare_step(*batches[0])
                for k in range(3):
                    nxt = tr3.prepare_step(*batches[k + 1]) if k < 2 else None
                    seen.append((float(tr3.training_step(*batches[k], ticket=tk)), tr3.last_neg.clone()))
                    tk = nxt
            else:
                for k in range(3):
                    seen.append((float(tr3.training_step(*batches[k])), tr3.last_neg.clone()))
            tbl3.check_overflow()
            runs.append((tbl3.item_local.clone(), tower3.weight.detach().clone(), seen))
        for (la, na), (lb, nb) in zip(runs[0][2], runs[1][2]):
            assert torch.equal(na, nb)                                       # the same draws
            np.testing.assert_allclose(la, lb, rtol=1e-5)
        np.testing.assert_allclose(runs[1][0].cpu(), runs[0][0].cpu(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(runs[1][1].cpu(), runs[0][1].cpu(), rtol=1e-5, atol=1e-7)
        assert (runs[0][0] - plan.take(item_d, rank)).abs().max() > 1e-4     # the steps trained
        # the forward step one batch ahead (prepare_forward on the second stream / ticket), whole and in two slices
        for ch in (1, 2):
            plain = ShardedItemTable(plan.take(item_d, rank).contiguous(), plan, rank, comm, sample_seed=41, chunks=ch)
            ahead = ShardedItemTable(plan.take(item_d, rank).contiguous(), plan, rank, comm, sample_seed=41, chunks=ch)
            smp = ra.UniformSampler(N)
            b256 = [(uid[:256].roll(k).contiguous(), pos[:256].roll(k).contiguous()) for k in range(3)]
            tk = ahead.prepare_forward(b256[0][1], n, smp, fused_loss='bpr')
            for k in range(3):
                nxt = ahead.prepare_forward(b256[k + 1][1], n, smp, fused_loss='bpr') if k < 2 else None
                a = plain.sample_and_score(user, *b256[k], n, smp, fused_loss='bpr', want_grad=True)
                b = ahead.sample_and_score(user, *b256[k], n, smp, fused_loss='bpr', want_grad=True, ticket=tk)
                tk = nxt
                assert torch.equal(a['neg_ids'], b['neg_ids'])
                assert torch.equal(a['pos_score'], b['pos_score']) and torch.equal(a['neg_score'], b['neg_score'])
                np.testing.assert_allclose(float(a['loss']), float(b['loss']), rtol=1e-6)
            ahead.check_overflow()
        open(os.path.join(result_dir, f'ok{rank}'), 'w').write('ok')
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('layout', ['block', 'interleaved'])
def test_two_ranks_on_one_gpu_hip_backend(tmp_path, layout):
    """World size 2 with the HIP backend on the single test GPU (collectives staged over gloo): forward with exact and
    fixed-capacity exchange, G-invariant negatives, gradient exchange, sharded full-catalog pass, training step -- for
    contiguous row blocks and for interleaved rows (RowShardPlan(layout=...))."""
    import torch.multiprocessing as mp
    mp.spawn(_two_rank_worker, args=(2, _free_port(), 'gloo', str(tmp_path), layout), nprocs=2, join=True)
    assert all(os.path.exists(tmp_path / f'ok{r}') for r in range(2))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs')
def test_two_gpus_rccl(tmp_path):
    """The same worker over RCCL, one rank per GPU (all_gather_into_tensor / all_to_all_single with and without
    split sizes / reduce_scatter_tensor / all_reduce on device buffers)."""
    import torch.multiprocessing as mp
    mp.spawn(_two_rank_worker, args=(2, _free_port(), 'nccl', str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(tmp_path / f'ok{r}') for r in range(2))


def test_world1_rccl_step_equals_unsharded():
    import torch.distributed as dist
    import recstudio_amd as ra
    from recstudio_amd.shard import RowShardPlan, ShardedItemTable
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        N, U, d, B, n = 30011, 500, 128, 257, 64
        g = torch.Generator().manual_seed(3)
        item = (torch.randn(N, d, generator=g) * 0.1).to(DEV)
        user = (torch.randn(U, d, generator=g) * 0.1).to(DEV)
        uid = torch.randint(1, U, (B,), generator=g).to(DEV)
        pos = torch.randint(1, N, (B,), generator=g).to(DEV)
        counts = (torch.rand(N, generator=g) ** 3 * 100).long()
        for sampler in (ra.UniformSampler(N), ra.PopularSamplerModel(counts).to(DEV)):
            # the table draws from its own job-wide stream (seed sample_seed, offset 0 on a fresh table)
            table = ShardedItemTable(item, RowShardPlan(N, 1), 0, dist, sample_seed=11, force_collectives=True)
            out = table.sample_and_score(user, uid, pos, n, sampler)         # exact split + calibration
            torch.manual_seed(11)
            score, ids = ra.retriever_scores(item, user, n, query_index=uid, pos_ids=pos, sampler=sampler)
            assert torch.equal(out['neg_ids'], ids)                     # same Philox stream either way
            np.testing.assert_allclose(out['neg_score'].cpu(), score['neg_score'].cpu(), rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(out['pos_score'].cpu(), score['pos_score'].cpu(), rtol=1e-5, atol=1e-6)
            want_p, want_n = oracle.retriever_forward(item.cpu(), user.cpu()[uid.cpu()], pos.cpu(), ids.cpu())
            np.testing.assert_allclose(out['neg_score'].cpu(), want_n, rtol=1e-4, atol=1e-6)
            # gradient exchange == the unsharded backward (this step runs the fixed-capacity exchange)
            out = table.sample_and_score(user, uid, pos, n, sampler, keep_route=True)
            assert 'send_counts' not in out['route']
            want_p, want_n = oracle.retriever_forward(item.cpu(), user.cpu()[uid.cpu()], pos.cpu(), out['neg_ids'].cpu())
            np.testing.assert_allclose(out['neg_score'].cpu(), want_n, rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(out['pos_score'].cpu(), want_p, rtol=1e-4, atol=1e-6)
            loss, dpos, dneg, _ = ra.ops.pairwise_loss(ra._native.LOSS_BPR, out['pos_score'], out['neg_score'])
            ig = torch.zeros_like(item)
            qg = table.backward(out['route'], dpos, dneg, ig)
            ig2, _, qg2 = ra.ops.fused_backward(item, user, out['neg_ids'], dneg, query_index=uid, pos_ids=pos, dpos=dpos)
            np.testing.assert_allclose(qg.cpu(), qg2.cpu(), rtol=2e-4, atol=1e-8)
            np.testing.assert_allclose(ig.cpu(), ig2.cpu(), rtol=2e-4, atol=1e-8)
        # query slices with asynchronous RCCL exchanges (real work objects on the communicator's stream)
        _check_pipelined_equals_whole(ra, ShardedItemTable,
                                      lambda chunks: ShardedItemTable(item, RowShardPlan(N, 1), 0, dist, sample_seed=5,
                                                                      chunks=chunks, force_collectives=True),
                                      user, uid[:256].contiguous(), pos[:256].contiguous(), 128,
                                      ra.PopularSamplerModel(counts).to(DEV), N, d, DEV)
    finally:
        dist.destroy_process_group()


def test_full_partials_of_two_shards_merge_to_the_unsharded_result():
    """HipBackend.full_partial on the two row blocks of a 2-way plan (block 1 has no padding row: base
    pointer trick) + merge kernels == rsa_fullscore over the whole table."""
    import recstudio_amd as ra
    from recstudio_amd.shard import HipBackend, RowShardPlan
    torch.manual_seed(0)
    N, d, B, k = 40_001, 128, 130, 50
    item = torch.randn(N, d, device=DEV) * 0.2
    item[0] = 0
    q = torch.randn(B, d, device=DEV) * 0.2
    plan = RowShardPlan(N, 2)
    hb = HipBackend()
    lses, vals, ids = [], [], []
    for r in range(2):
        lo, hi = plan.bounds(r)
        lse, tv, ti = hb.full_partial(item[lo:hi].contiguous(), q, k, True, r == 0)
        lses.append(lse)
        vals.append(tv)
        ids.append(ti + (lo if r == 0 else lo - 1))
    lse = hb.merge_lse(torch.stack(lses, 1))
    tv, ti = hb.merge_topk(torch.cat(vals, 1), torch.cat(ids, 1), k)
    _, wl, wv, wi = ra.ops.fullscore(item, q, want_lse=True, k=k)
    assert torch.equal(ti, wi)
    np.testing.assert_allclose(tv.cpu(), wv.cpu(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(lse.cpu(), wl.cpu(), rtol=1e-6)
    sc = q.double() @ item[1:].double().t()
    np.testing.assert_allclose(lse.cpu(), torch.logsumexp(sc, -1).cpu(), rtol=1e-5)
    # fp64 ordering can swap fp32 near-ties: compare the values, and the ids where they agree on rank
    w64 = torch.topk(sc, k)
    np.testing.assert_allclose(tv.cpu(), w64.values.float().cpu(), rtol=1e-5, atol=1e-6)
    assert (ti == w64.indices + 1).float().mean() > 0.99


def test_world1_rccl_full_catalog_pass():
    import torch.distributed as dist
    import recstudio_amd as ra
    from recstudio_amd.shard import RowShardPlan, ShardedItemTable
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        torch.manual_seed(1)
        N, d, B, k = 50_001, 64, 70, 20
        item = torch.randn(N, d, device=DEV) * 0.2
        q = torch.randn(B, d, device=DEV) * 0.2
        table = ShardedItemTable(item, RowShardPlan(N, 1), 0, dist, force_collectives=True)
        lse, tv, ti = table.full_lse_topk(q, k)
        _, wl, wv, wi = ra.ops.fullscore(item, q, want_lse=True, k=k)
        assert torch.equal(ti, wi) and torch.equal(tv, wv)
        np.testing.assert_allclose(lse.cpu(), wl.cpu(), rtol=1e-6)
    finally:
        dist.destroy_process_group()


def test_world1_rccl_sharded_training_step():
    """ShardedRetriever.training_step through the HIP kernels (world_size 1 over RCCL) == torch autograd of
    the same BPR loss with the same sampled negatives."""
    import torch.distributed as dist
    import recstudio_amd as ra
    from recstudio_amd.shard import RowShardPlan, ShardedItemTable, ShardedRetriever
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        torch.manual_seed(2)
        N, U, d, B, n = 20_011, 300, 128, 200, 16
        item = torch.randn(N, d, device=DEV) * 0.2
        item[0] = 0
        tower = torch.nn.Embedding(U, d).to(DEV)
        uid = torch.randint(1, U, (B,), device=DEV)
        pos = torch.randint(1, N, (B,), device=DEV)
        table = ShardedItemTable(item, RowShardPlan(N, 1), 0, dist, force_collectives=True)
        trainer = ShardedRetriever(table, tower, ra.UniformSampler(N), ra.BPRLoss(), n, sparse_query_rows=True, keep_neg_ids=True)
        loss = trainer.training_step(uid, pos)
        neg = trainer.last_neg
        item_ref = item.clone().requires_grad_(True)
        w_ref = tower.weight.detach().clone().requires_grad_(True)
        q = w_ref[uid]
        ps = (q * item_ref[pos]).sum(-1)
        ns = (q.unsqueeze(1) * item_ref[neg]).sum(-1)
        ref = -torch.nn.functional.logsigmoid(ps.view(-1, 1) - ns).mean(-1).mean()
        ref.backward()
        np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-5)
        want = item_ref.grad.clone()
        want[0] = 0
        np.testing.assert_allclose(trainer.item_grad_local.cpu(), want.cpu(), rtol=2e-4, atol=1e-7)
        assert tower.weight.grad is None               # an nn.Embedding tower's gradient stays row-sparse
        np.testing.assert_allclose(trainer.query_grad_dense().cpu(), w_ref.grad.cpu(), rtol=2e-4, atol=1e-7)
        # the dense (autograd + all-reduce) path on request
        tower.weight.grad = None
        table_d = ShardedItemTable(item, RowShardPlan(N, 1), 0, dist)
        dense = ShardedRetriever(table_d, tower, ra.UniformSampler(N), ra.BPRLoss(), n)
        assert not dense.sparse_query_rows                   # the default: autograd fills .grad, summed over the ranks
        torch.manual_seed(77)
        dense.training_step(uid, pos)
        g_fused = tower.weight.grad.clone()
        assert torch.isfinite(g_fused).all() and g_fused.abs().sum() > 0
        # a loss plugin the home kernel does not know runs under autograd over the exchanged scores: same negatives
        # (same job-wide stream state), same gradient for the same loss
        class MyBPR(ra.BPRLoss):
            pass
        tower.weight.grad = None
        table_p = ShardedItemTable(item, RowShardPlan(N, 1), 0, dist)
        plug = ShardedRetriever(table_p, tower, ra.UniformSampler(N), MyBPR(), n)
        assert plug._fused_loss_kind() is None
        plug.training_step(uid, pos)
        np.testing.assert_allclose(tower.weight.grad.cpu(), g_fused.cpu(), rtol=2e-4, atol=1e-8)
        np.testing.assert_allclose(plug.item_grad_local.cpu(), dense.item_grad_local.cpu(), rtol=2e-4, atol=1e-8)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('kind,n', [('uniform', 64), ('popular', 64), ('popular', 256), ('uniform', 16)])
def test_world1_rccl_sampled_softmax_on_owners(kind, n):
    """Stock SampledSoftmaxLoss through ShardedRetriever with the loss evaluated ON THE OWNERS (rsa_shard_owner_ssm_forward /
    _finish: one walk over the received rows leaves per-query (max, sum, sum * row) partials, an 8-byte-per-query all-reduce,
    then d per slot + the sorted apply pass) == torch autograd of loss_func.py:80-90 with the same negatives: loss, the dense
    item gradient block, the row-sparse query gradient; the in-place SGD form == the gradient form; == the score-at-home
    protocol (home kernel) of the same step."""
    import torch.distributed as dist
    import recstudio_amd as ra
    from recstudio_amd.shard import RowShardPlan, ShardedItemTable, ShardedRetriever
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        torch.manual_seed(4)
        N, U, d, B = 20_011, 300, 128, 300
        item = torch.randn(N, d, device=DEV) * 0.2
        item[0] = 0
        tower = torch.nn.Embedding(U, d).to(DEV)
        uid = torch.randint(1, U, (B,), device=DEV)
        pos = torch.randint(1, N, (B,), device=DEV)
        if kind == 'uniform':
            smp = ra.UniformSampler(N)
        else:
            smp = ra.PopularSamplerModel((torch.rand(N) ** 4 * 300).long() + 1).to(DEV)

        def run(**kw):
            table = ShardedItemTable(item.clone(), RowShardPlan(N, 1), 0, dist, force_collectives=True, sample_seed=9)
            tr = ShardedRetriever(table, tower, smp, ra.SampledSoftmaxLoss(), n, sparse_query_rows=True, keep_neg_ids=True, **kw)
            return tr, tr.training_step(uid, pos)
        trainer, loss = run()
        assert trainer.table.ssm_owner_ok()
        neg = trainer.last_neg
        item_ref = item.clone().requires_grad_(True)
        w_ref = tower.weight.detach().clone().requires_grad_(True)
        q = w_ref[uid]
        zp = (q * item_ref[pos]).sum(-1)
        zn = (q.unsqueeze(1) * item_ref[neg]).sum(-1)
        if kind == 'popular':
            zp = zp - torch.log(smp.pop_prob[pos])
            zn = zn - torch.log(smp.pop_prob[neg])
        ref = (torch.logsumexp(torch.cat([zp.view(-1, 1), zn], 1), -1) - zp).mean()
        ref.backward()
        np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-5)
        want = item_ref.grad.clone()
        want[0] = 0
        np.testing.assert_allclose(trainer.item_grad_local.cpu(), want.cpu(), rtol=3e-4, atol=1e-7)
        np.testing.assert_allclose(trainer.query_grad_dense().cpu(), w_ref.grad.cpu(), rtol=3e-4, atol=1e-7)
        # the score-at-home protocol of the same step: the same draws (same job-wide stream), the same numbers
        home, loss_h = run(owner_ssm=False)
        assert torch.equal(home.last_neg, neg)
        np.testing.assert_allclose(loss_h.item(), loss.item(), rtol=1e-5)
        np.testing.assert_allclose(home.item_grad_local.cpu(), trainer.item_grad_local.cpu(), rtol=3e-4, atol=1e-7)
        # plain SGD in place inside the owners' pass == the gradient block applied by hand
        sgd, _ = run(item_sgd_lr=0.3)
        np.testing.assert_allclose(sgd.table.item_local.cpu(), (item - 0.3 * trainer.item_grad_local).cpu(), rtol=1e-5, atol=1e-7)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('loss_name', ['bpr', 'ssm'])
@pytest.mark.parametrize('kind', ['uniform', 'popular'])
def test_deterministic_router_makes_sharded_steps_bit_reproducible(kind, loss_name):
    """ShardedItemTable(deterministic=True): the router's slots come from a count pass + a prefix over the workgroups instead of
    atomic cursors, so three in-place SGD training steps (BPR and SampledSoftmax on the owners) leave bit-identical item rows,
    user rows, losses and send buffers run to run -- and the SAME numbers (to fp32 rounding: another summation order) as the
    default router.  The draws are those of the default router bit for bit."""
    import torch.distributed as dist
    import recstudio_amd as ra
    from recstudio_amd.shard import RowShardPlan, ShardedItemTable, ShardedRetriever
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        torch.manual_seed(6)
        N, U, d, B, n = 40_009, 700, 128, 1024, 64
        item = torch.randn(N, d, device=DEV) * 0.2
        item[0] = 0
        user = torch.randn(U, d, device=DEV) * 0.2
        batches = [(torch.randint(1, U, (B,), device=DEV), torch.randint(1, N, (B,), device=DEV)) for _ in range(3)]
        smp = ra.UniformSampler(N) if kind == 'uniform' else ra.PopularSamplerModel((torch.rand(N) ** 4 * 300).long() + 1).to(DEV)
        loss_fn = ra.BPRLoss() if loss_name == 'bpr' else ra.SampledSoftmaxLoss()

        def run(det):
            tower = torch.nn.Embedding(U, d).to(DEV)
            with torch.no_grad():
                tower.weight.copy_(user)
            table = ShardedItemTable(item.clone(), RowShardPlan(N, 1), 0, dist, force_collectives=True, sample_seed=13,
                                     deterministic=det)
            tr = ShardedRetriever(table, tower, smp, loss_fn, n, item_sgd_lr=0.4, query_sgd_lr=0.2, keep_neg_ids=True)
            seen = [(tr.training_step(u, p).clone(), tr.last_neg.clone()) for u, p in batches]
            table.check_overflow()
            return table.item_local.clone(), tower.weight.detach().clone(), seen
        a, b, c = run(True), run(True), run(False)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        for (la, na), (lb, nb), (lc, nc) in zip(a[2], b[2], c[2]):
            assert torch.equal(la, lb) and torch.equal(na, nb) and torch.equal(na, nc)
            np.testing.assert_allclose(la.item(), lc.item(), rtol=1e-5)
        np.testing.assert_allclose(a[0].cpu(), c[0].cpu(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(a[1].cpu(), c[1].cpu(), rtol=1e-4, atol=1e-6)
        assert (a[0] - item).abs().max() > 1e-4
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('layout', ['block', 'interleaved'])
def test_launch_entry_point_world1_loss_decreases(layout):
    """recstudio_amd.launch (the multi-GPU training entry point) as a single rank: runs, and SGD on the sharded
    step lowers the BPR loss on a tiny synthetic problem."""
    from recstudio_amd import launch
    os.environ['MASTER_PORT'] = str(_free_port())
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        os.environ.pop(k, None)
    losses = launch.main(['--items', '2001', '--users', '301', '--dim', '64', '--neg', '16', '--batch', '512',
                          '--steps', '61', '--lr', '100.0', '--layout', layout])
    assert losses[0] == pytest.approx(0.6931, abs=2e-3) and losses[-1] < losses[0] - 0.05


def test_world1_rccl_inplace_item_sgd_equals_dense_gradient_step():
    """ShardedRetriever(item_sgd_lr=lr): the item rows updated inside the backward exchange == weights minus lr
    times the dense item gradient of the same step (same seed, hence the same negatives)."""
    import torch.distributed as dist
    import recstudio_amd as ra
    from recstudio_amd.shard import RowShardPlan, ShardedItemTable, ShardedRetriever
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        torch.manual_seed(4)
        N, U, d, B, n, lr = 9001, 200, 64, 150, 8, 0.3
        item0 = torch.randn(N, d, device=DEV) * 0.2
        item0[0] = 0
        uid = torch.randint(1, U, (B,), device=DEV)
        pos = torch.randint(1, N, (B,), device=DEV)
        results = []
        for inplace in (False, True):
            item = item0.clone()
            tower = torch.nn.Embedding(U, d).to(DEV)
            with torch.no_grad():
                tower.weight.copy_(torch.linspace(-1, 1, U * d, device=DEV).view(U, d))
            table = ShardedItemTable(item, RowShardPlan(N, 1), 0, dist, force_collectives=inplace)
            trainer = ShardedRetriever(table, tower, ra.UniformSampler(N), ra.BPRLoss(), n, item_sgd_lr=lr if inplace else None,
                                       sparse_query_rows=True)
            torch.manual_seed(99)
            trainer.training_step(uid, pos)
            results.append(item if inplace else item - lr * trainer.item_grad_local)
            results.append(trainer.query_grad_dense())
        np.testing.assert_allclose(results[2].cpu(), results[0].cpu(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(results[3].cpu(), results[1].cpu(), rtol=1e-5, atol=1e-7)
    finally:
        dist.destroy_process_group()


def test_world1_rccl_overflow_step_is_harmless():
    """A step whose ids outgrow the calibrated capacity: the dropped elements have no score, no
    loss term and no gradient -- the loss stays finite -- and because every rank learns from the segment headers that
    something was dropped, the step's in-place SGD updates (item rows AND user rows) are scaled by 0 on the device:
    the weights are bit-identical afterwards.  The sticky count then makes check_overflow raise, and the next step
    recalibrates and trains again."""
    import torch.distributed as dist
    import recstudio_amd as ra
    from recstudio_amd.shard import RowShardPlan, ShardedItemTable, ShardedRetriever
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        torch.manual_seed(5)
        N, U, d, B, n = 20_011, 300, 128, 256, 64
        item = torch.randn(N, d, device=DEV) * 0.2
        item[0] = 0
        tower = torch.nn.Embedding(U, d).to(DEV)
        uid = torch.randint(1, U, (B,), device=DEV)
        pos = torch.randint(1, N, (B,), device=DEV)
        for loss_fn, own in ((ra.BPRLoss(), True), (ra.SampledSoftmaxLoss(), True), (ra.SampledSoftmaxLoss(), False)):
            table = ShardedItemTable(item, RowShardPlan(N, 1), 0, dist, check_every=0)
            trainer = ShardedRetriever(table, tower, ra.UniformSampler(N), loss_fn, n, item_sgd_lr=0.3, query_sgd_lr=0.3,
                                       owner_ssm=own)
            l0 = trainer.training_step(uid, pos)                           # calibrates, trains
            assert torch.isfinite(l0) and int(table.state['step_dropped']) == 0
            key = (B, n, 1, 'owners') if own else (B, n, 1)               # (loss on the owners / scores sent home)
            table._cap[key] = table._cap[key] // 2                         # force an overflow: half the elements fit
            w_item, w_user = item.clone(), tower.weight.detach().clone()
            l1 = trainer.training_step(uid, pos)
            assert torch.isfinite(l1) and float(l1) > 0
            assert 0 < int(table.state['step_dropped']) < B * (n + 1)
            assert torch.equal(item, w_item) and torch.equal(tower.weight.detach(), w_user)      # untouched
            with pytest.raises(RuntimeError, match='did not fit'):
                table.check_overflow()
            l2 = trainer.training_step(uid, pos)                           # recalibrated: a normal step again
            assert int(table.state['step_dropped']) == 0 and not torch.equal(item, w_item)
            table.check_overflow()
    finally:
        dist.destroy_process_group()


def _fit_gpu_worker(rank, world, port, result_dir, layout='block', train_extra=None):
    import torch.distributed as dist
    import recstudio_amd as ra
    from recstudio_amd.dataset import TripletDataset
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'data_ml100k.npz'))
        conf = {'train': {'epochs': 2, 'batch_size': 2048 // world, 'negative_count': 64, 'seed': 2022, 'learning_rate': 0.01,
                          'early_stop_patience': 100, 'shard_layout': layout},
                'eval': {'batch_size': 128 // world, 'cutoff': [10], 'val_metrics': ['ndcg', 'recall'], 'topk': 50,
                         'test_metrics': ['ndcg', 'recall']},
                'model': {'embed_dim': 64}}
        conf['train'].update(train_extra or {})
        model = ra.BPR(conf)
        ds = TripletDataset('ml-100k', {'low_rating_thres': 3.0},
                            _interactions=(g['raw_user'].astype(str), g['raw_item'].astype(str),
                                           g['raw_rating'].astype(np.float64), g['raw_time'].astype(np.float64)))
        trn, val, tst = ds.build(split_ratio=[0.8, 0.1, 0.1], shuffle=True, split_mode='user_entry')
        torch.cuda.manual_seed_all(2022)               # the device loader's permutation comes from the device generator
        best = model.fit(trn, val, dist=StagedDist(dist), device='cuda:0')
        test = model.evaluate(tst, verbose=False)
        torch.save({'best': best, 'val': dict(model.logged_metrics), 'test': test, 'losses': torch.cat(model.train_losses),
                    'item': model.item_encoder.weight.detach().cpu(), 'lo': model._shard['lo'],
                    'tower': model.query_encoder.weight.detach().cpu(), 'lookahead': bool(model._shard.get('lookahead'))},
                   os.path.join(result_dir, f'w{world}r{rank}.pt'))
    finally:
        dist.destroy_process_group()


def test_fit_two_staged_ranks_equals_one_rank_hip(tmp_path):
    """``BaseRetriever.fit`` / ``evaluate`` with the HIP backend as two ranks on the test GPU (collectives staged over gloo;
    over RCCL the same code runs one rank per GPU): ml-100k BPR, n = 64 drawn inside the routing launch, fused BPR home
    kernel, dense Adam on the row blocks and the replicated tower, device loader with rank parts, sharded top-k with
    history masking -- losses, ndcg@10 / recall@10 and weights equal the one-rank run of the same global batch."""
    import torch.multiprocessing as mp
    for world in (1, 2):
        mp.spawn(_fit_gpu_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    one = torch.load(tmp_path / 'w1r0.pt', weights_only=False)
    two = [torch.load(tmp_path / f'w2r{r}.pt', weights_only=False) for r in range(2)]
    assert abs(float(one['losses'][0]) - 0.6931) < 2e-3 and float(one['losses'][-1]) < float(one['losses'][0]) - 0.05
    for t in two:
        np.testing.assert_allclose(t['losses'].numpy(), one['losses'].numpy(), rtol=2e-5, atol=1e-6)
        for k in ('ndcg@10', 'recall@10'):
            assert abs(t['val'][k] - one['val'][k]) < 1e-4 and abs(t['test'][k] - one['test'][k]) < 1e-4
        np.testing.assert_allclose(t['tower'].numpy(), one['tower'].numpy(), rtol=1e-3, atol=1e-5)
    items = torch.cat([two[0]['item'], two[1]['item']])
    np.testing.assert_allclose(items.numpy(), one['item'].numpy(), rtol=1e-3, atol=1e-5)
    assert one['val']['ndcg@10'] > 0.01 and not items[0].any()
    # train.shard_layout 'interleaved': rank r trains rows r, r + 2, ... of the same table
    os.makedirs(tmp_path / 'il')
    mp.spawn(_fit_gpu_worker, args=(2, _free_port(), str(tmp_path / 'il'), 'interleaved'), nprocs=2, join=True)
    il = [torch.load(tmp_path / 'il' / f'w2r{r}.pt', weights_only=False) for r in range(2)]
    for t in il:
        np.testing.assert_allclose(t['losses'].numpy(), one['losses'].numpy(), rtol=2e-5, atol=1e-6)
        for k in ('ndcg@10', 'recall@10'):
            assert abs(t['val'][k] - one['val'][k]) < 1e-4 and abs(t['test'][k] - one['test'][k]) < 1e-4
    items = torch.empty_like(one['item'])
    items[0::2], items[1::2] = il[0]['item'], il[1]['item']
    np.testing.assert_allclose(items.numpy(), one['item'].numpy(), rtol=1e-3, atol=1e-5)


def test_fit_two_staged_ranks_one_batch_ahead_hip(tmp_path):
    """``train.shard_lookahead`` through ``BaseRetriever.fit`` with the HIP backend (two staged ranks, in-kernel SGD so that the
    step is the owner-side BPR step with in-place updates): every step's negatives / routing / key exchange / owner sorts
    issued one batch ahead on the second stream, the epoch loop on a high-priority stream -- the same run as without."""
    import torch.multiprocessing as mp
    extra = {'fused_optimizer': 'sgd', 'learning_rate': 200.0, 'epochs': 1}
    runs = {}
    for tag, ex in (('plain', extra), ('ahead', dict(extra, shard_lookahead=True))):
        os.makedirs(tmp_path / tag)
        mp.spawn(_fit_gpu_worker, args=(2, _free_port(), str(tmp_path / tag), 'block', ex), nprocs=2, join=True)
        runs[tag] = [torch.load(tmp_path / tag / f'w2r{r}.pt', weights_only=False) for r in range(2)]
    for a, b in zip(runs['plain'], runs['ahead']):
        assert b['lookahead'] and not a['lookahead']
        np.testing.assert_allclose(b['losses'].numpy(), a['losses'].numpy(), rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(b['item'].numpy(), a['item'].numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(b['tower'].numpy(), a['tower'].numpy(), rtol=1e-4, atol=1e-6)
    assert float(runs['plain'][0]['losses'][-1]) < float(runs['plain'][0]['losses'][0]) - 2e-4      # the epoch trained


@pytest.mark.parametrize('d', [32, 128])
def test_apply_rows_pad_row_semantics_do_not_depend_on_dim(d):
    """HipBackend.apply_rows(pad_row=-1) updates EVERY row, row 0 included, for the stock dims (sorted scatter) and
    for the others alike; pad_row=0 skips row 0 in both."""
    from recstudio_amd.shard import HipBackend
    torch.manual_seed(d)
    U, m = 50, 400
    ids = torch.randint(0, U, (m,), device=DEV)
    ids[:7] = 0
    rows = torch.randn(m, d, device=DEV)
    for pad in (-1, 0):
        table = torch.zeros(U, d, device=DEV)
        HipBackend().apply_rows(table, ids, rows, -0.5, pad_row=pad)
        keep = ids != pad
        want = torch.zeros(U, d, device=DEV).index_add_(0, ids[keep], -0.5 * rows[keep])
        np.testing.assert_allclose(table.cpu(), want.cpu(), rtol=1e-5, atol=1e-6)
        assert bool(table[0].any()) == (pad == -1)
