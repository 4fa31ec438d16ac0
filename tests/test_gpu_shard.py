"""GPU: the HIP routing kernels of the sharded path against the checker backend used by the gloo
test, and a world_size-1 RCCL run of the whole sharded step against the unsharded fused forward."""
import os

import numpy as np
import pytest
import torch

import oracle
from test_shard_gloo import CheckerBackend, _free_port

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('world,B,n', [(4, 33, 7), (8, 1000, 64), (2, 5, 1), (3, 17, 100)])
def test_route_kernels_match_checker(world, B, n):
    from recstudio_amd.shard import HipBackend, RowShardPlan
    n_items = 10007
    plan = RowShardPlan(n_items, world)
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
            table = ShardedItemTable(item, RowShardPlan(N, 1), 0, dist)
            torch.manual_seed(11)
            out = table.sample_and_score(user, uid, pos, n, sampler)
            torch.manual_seed(11)
            score, ids = ra.retriever_scores(item, user, n, query_index=uid, pos_ids=pos, sampler=sampler)
            assert torch.equal(out['neg_ids'], ids)                     # same Philox stream either way
            np.testing.assert_allclose(out['neg_score'].cpu(), score['neg_score'].cpu(), rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(out['pos_score'].cpu(), score['pos_score'].cpu(), rtol=1e-5, atol=1e-6)
            want_p, want_n = oracle.retriever_forward(item.cpu(), user.cpu()[uid.cpu()], pos.cpu(), ids.cpu())
            np.testing.assert_allclose(out['neg_score'].cpu(), want_n, rtol=1e-4, atol=1e-6)
            # gradient exchange == the unsharded backward
            torch.manual_seed(11)
            out = table.sample_and_score(user, uid, pos, n, sampler, keep_route=True)
            loss, dpos, dneg, _ = ra.ops.pairwise_loss(ra._native.LOSS_BPR, out['pos_score'], out['neg_score'])
            ig = torch.zeros_like(item)
            qg = table.backward(out['route'], dpos, dneg, ig)
            ig2, _, qg2 = ra.ops.fused_backward(item, user, out['neg_ids'], dneg, query_index=uid, pos_ids=pos, dpos=dpos)
            np.testing.assert_allclose(qg.cpu(), qg2.cpu(), rtol=2e-4, atol=1e-8)
            np.testing.assert_allclose(ig.cpu(), ig2.cpu(), rtol=2e-4, atol=1e-8)
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
        table = ShardedItemTable(item, RowShardPlan(N, 1), 0, dist)
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
        table = ShardedItemTable(item, RowShardPlan(N, 1), 0, dist)
        trainer = ShardedRetriever(table, tower, ra.UniformSampler(N), ra.BPRLoss(), n)
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
        np.testing.assert_allclose(tower.weight.grad.cpu(), w_ref.grad.cpu(), rtol=2e-4, atol=1e-7)
    finally:
        dist.destroy_process_group()


def test_launch_entry_point_world1_loss_decreases():
    """recstudio_amd.launch (the multi-GPU training entry point) as a single rank: runs, and SGD on the sharded
    step lowers the BPR loss on a tiny synthetic problem."""
    from recstudio_amd import launch
    os.environ['MASTER_PORT'] = str(_free_port())
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        os.environ.pop(k, None)
    losses = launch.main(['--items', '2001', '--users', '301', '--dim', '64', '--neg', '16', '--batch', '512',
                          '--steps', '61', '--lr', '100.0'])
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
            table = ShardedItemTable(item, RowShardPlan(N, 1), 0, dist)
            trainer = ShardedRetriever(table, tower, ra.UniformSampler(N), ra.BPRLoss(), n, item_sgd_lr=lr if inplace else None)
            torch.manual_seed(99)
            trainer.training_step(uid, pos)
            results.append(item if inplace else item - lr * trainer.item_grad_local)
            results.append(tower.weight.grad.clone())
        np.testing.assert_allclose(results[2].cpu(), results[0].cpu(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(results[3].cpu(), results[1].cpu(), rtol=1e-5, atol=1e-7)
    finally:
        dist.destroy_process_group()
