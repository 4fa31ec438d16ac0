"""GPU tests added in round 4 (run with ``-m gpu`` on an MI355X): parity at the north-star's own size (a 10^8-item,
51 GB table on one GPU), the accuracy anchor against the reference's published ml-100k run, item-tower models under
the sharded fit with the HIP backend, and the kernels this round added."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu

DEV = 'cuda'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'tools'))


@pytest.fixture(scope='module')
def ra():
    import recstudio_amd
    recstudio_amd._native.lib()          # fail loudly if the HIP extension is not there
    assert torch.cuda.is_available()
    torch.cuda.init()
    return recstudio_amd


def rel_close(a, b, rtol=1e-4, atol=1e-6):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


# --------------------------------------------------------------------------- north_star: d = 128, a 100 M-item table
def test_north_star_size_1e8_items(ra):
    """BASELINE.json's target configuration on ONE GPU: N = 100 000 001 items x d = 128 fp32 (51.2 GB), the sizes the
    bench's ``table_100M`` figures are timed on.  Reference semantics: sampler.py:102-104 (uniform), :246-258 (popularity),
    baseretriever.py:153-171 (gather + score), loss_func.py:55-59 (BPR).
      (1) uniform, n = 64, B = 65 536: ids == torch.randint on this device, bit for bit;
      (2) popularity (g = 21 bucket lines, 268 MB): ids == searchsorted(table, rand), log-probs == log(pop_prob[ids]);
      (3) rows whose byte offset is beyond 4 GiB, beyond 32 GiB and the LAST row, as given ids -> scores == oracle;
      (4) oracle spot check of 2000 sampled elements; idempotence (sampled ids re-scored as given ids: same bits);
      (5) n = 1024, B = 4096 (configs[3] per-GPU shape, the walk kernel): loss / d loss/d score == the separate loss
          kernel on the same scores, == oracle BPR on a subset of queries, and bit-equal run to run."""
    N, U, d = 100_000_001, 200_001, 128
    if torch.cuda.get_device_properties(0).total_memory < 80e9:
        pytest.skip('needs ~60 GB of device memory')
    g = torch.Generator(device=DEV).manual_seed(4)
    iw = torch.empty(N, d, device=DEV)
    for lo in range(0, N, 10_000_000):                # (normal_ on the whole 51 GB tensor needs no temporaries either; chunks keep it obvious)
        iw[lo:lo + 10_000_000].normal_(0, 0.02, generator=g)
    iw[0] = 0
    uw = torch.empty(U, d, device=DEV).normal_(0, 0.02, generator=g)
    B, n = 65_536, 64
    uid = torch.randint(1, U, (B,), device=DEV, generator=g)
    pos = torch.randint(1, N, (B,), device=DEV, generator=g)
    pos[:4] = torch.tensor([N - 1, 8_388_609, 67_108_865, 1], device=DEV)
    # (1) uniform sampler
    us = ra.UniformSampler(N)
    torch.manual_seed(2022)
    s_u, ids_u = ra.retriever_scores(iw, uw, n, query_index=uid, pos_ids=pos, sampler=us)
    torch.manual_seed(2022)
    assert torch.equal(ids_u, torch.randint(1, N, (B, n), device=DEV))
    assert int(ids_u.max()) > N - 1000 and int(ids_u.min()) >= 1        # the draw covers the table's far end
    assert s_u['log_neg_prob'].dtype == torch.int64 and not s_u['log_neg_prob'].any()
    # (2) popularity sampler: 28 % of the catalog never seen, a heavy head
    counts = (torch.rand(N, generator=torch.Generator().manual_seed(2)) ** 8 * 1e4).long()
    ps = ra.PopularSamplerModel(counts).to(DEV)
    assert ps.lookup_kwargs().get('cdf_lines') is not None and ps.lookup_kwargs()['lines_log2'] >= 20
    torch.manual_seed(7)
    s_p, ids_p = ra.retriever_scores(iw, uw, n, query_index=uid, pos_ids=pos, sampler=ps)
    torch.manual_seed(7)
    want = torch.searchsorted(ps.table, torch.rand(B, n, device=DEV)).clamp_(max=N - 1)
    assert torch.equal(ids_p, want)
    rel_close(s_p['log_neg_prob'].cpu(), torch.log(ps.pop_prob[ids_p]).cpu(), rtol=1e-6, atol=0)
    rel_close(s_p['log_pos_prob'].cpu(), torch.log(ps.pop_prob[pos]).cpu(), rtol=1e-6, atol=0)
    # (3) rows beyond 4 GiB / 32 GiB / the last row, as GIVEN ids
    far = torch.tensor([N - 1, N - 2, 8_388_608, 8_388_609, 67_108_864, 67_108_865, 99_999_999, 1], device=DEV)
    given = far.repeat(B, n // far.numel())
    s_g = ra.ops.fused_forward(iw, uw, n, query_index=uid, pos_ids=pos, neg_ids=given)
    q_head = uw[uid[:64]].cpu()
    rel_close(s_g['neg_score'][:64].cpu(), oracle.inner_product_score(q_head, iw[given[:64]].cpu()), rtol=1e-4, atol=1e-7)
    rel_close(s_g['pos_score'][:64].cpu(), oracle.inner_product_score(q_head, iw[pos[:64]].cpu()), rtol=1e-4, atol=1e-7)
    assert float(s_g['pos_score'][:3].abs().min()) > 0                   # (the far rows are not zeros read out of range)
    # (4) oracle spot check on 2000 sampled elements of both draws; idempotence
    gen = torch.Generator().manual_seed(5)
    sb, sj = torch.randint(0, B, (2000,), generator=gen), torch.randint(0, n, (2000,), generator=gen)
    for sc, ids in ((s_u, ids_u), (s_p, ids_p)):
        rows = iw[ids[sb.to(DEV), sj.to(DEV)]].cpu()
        rel_close(sc['neg_score'].cpu()[sb, sj], oracle.inner_product_score(uw[uid[sb.to(DEV)]].cpu(), rows), rtol=1e-4, atol=1e-7)
        again = ra.ops.fused_forward(iw, uw, n, query_index=uid, pos_ids=pos, neg_ids=ids)
        assert torch.equal(again['neg_score'], sc['neg_score']) and torch.equal(again['pos_score'], sc['pos_score'])
    del s_u, s_p, s_g, given, again
    # (5) the walk kernel at configs[3]'s per-GPU shape
    B2, n2 = 4096, 1024
    uid2, pos2 = uid[:B2].contiguous(), pos[:B2].contiguous()
    outs = []
    for rep in range(2):
        torch.manual_seed(11)
        outs.append(ra.ops.fused_forward(iw, uw, n2, query_index=uid2, pos_ids=pos2, sampler=ra._native.SAMPLER_UNIFORM,
                                         fused_bpr=True, want_query_grad=True))
    a, b = outs
    torch.manual_seed(11)
    assert torch.equal(a['neg_ids'], torch.randint(1, N, (B2, n2), device=DEV))
    for k in ('loss', 'row_loss', 'dpos', 'dneg', 'query_grad', 'neg_score', 'pos_score'):
        assert torch.equal(a[k], b[k]), k                                 # no atomics: bit-equal run to run
    loss2, dpos2, dneg2, row2 = ra.ops.pairwise_loss(ra._native.LOSS_BPR, a['pos_score'], a['neg_score'])
    rel_close(a['loss'].cpu(), loss2.cpu(), rtol=1e-6)
    rel_close(a['row_loss'].cpu(), row2.cpu(), rtol=1e-5, atol=1e-7)
    rel_close(a['dneg'].cpu(), dneg2.cpu(), rtol=1e-5, atol=1e-9)
    rel_close(a['dpos'].cpu(), dpos2.cpu(), rtol=1e-5, atol=1e-9)
    m = 32                                                                # oracle BPR + query gradient on the first queries
    qv = uw[uid2[:m]].cpu().requires_grad_(True)
    rows_n, rows_p = iw[a['neg_ids'][:m]].cpu(), iw[pos2[:m]].cpu()
    ps_o, ns_o = oracle.inner_product_score(qv, rows_p), oracle.inner_product_score(qv, rows_n)
    rel_close(a['neg_score'][:m].cpu(), ns_o.detach(), rtol=1e-4, atol=1e-7)
    per_row = -torch.nn.functional.logsigmoid(ps_o.view(-1, 1) - ns_o).mean(-1)
    rel_close(a['row_loss'][:m].cpu(), per_row.detach(), rtol=1e-5, atol=1e-7)
    (per_row.sum() / B2).backward()
    rel_close(a['query_grad'][:m].cpu(), qv.grad, rtol=2e-4, atol=1e-9)


# --------------------------------------------------------------------------- accuracy anchor (VERDICT r3 missing #4)
def _ml100k(ra, cls=None, **cfg):
    from test_dataset_golden import make
    g = np.load(os.path.join(HERE, 'golden', 'data_ml100k.npz'))
    return make(cls or ra.TripletDataset, g, **cfg)


def test_bpr_fit_ml100k_converges_to_reference(ra):
    """configs[0] end to end, the reference's stock configuration (README.md:125-181: BPR, d = 64, B = 512, n = 1, Adam
    1e-3, xavier_normal, seed 2022, early stopping on ndcg@5 with patience 10, eval batch 20, top-100):
      * the converged model lands where the reference does -- published (README.md:208): test ndcg@10 = 0.2442,
        recall@20 = 0.3530; the same reference code run in the build container (tests/golden/fit_bpr_ml100k.npz,
        oracle/make_golden_fit.py): 0.2424 / 0.3648 after 44 epochs;
      * the per-epoch training loss follows the reference's own trajectory (different random streams -- the reference
        shuffles and samples from the host generator --, same data, same optimizer): every one of the first 12 epochs
        within 2 %."""
    ref = np.load(os.path.join(HERE, 'golden', 'fit_bpr_ml100k.npz'))
    model = ra.BPR({'eval': {'batch_size': 20}})                      # every other key: the reference's defaults; seeds (2022) ...
    ds = _ml100k(ra)                                                  # ... then the data, quickstart's order (run.py:35, :56)
    trn, val, tst = ds.build(split_ratio=[0.8, 0.1, 0.1], shuffle=True, split_mode='user_entry')
    assert model.config['train']['learner'] == 'adam' and model.config['train']['negative_count'] == 1
    model.fit(trn, val)
    res = model.evaluate(tst, verbose=False)
    hist = model.history
    print('epochs', len(hist), 'test', {k: round(float(v), 4) for k, v in res.items() if k.endswith('@10') or k.endswith('@20')})
    assert 25 <= len(hist) <= 80                                      # early stopping worked (reference here: 44; published: 35)
    assert 0.22 <= res['ndcg@10'] <= 0.27 and 0.32 <= res['recall@20'] <= 0.39
    assert abs(res['ndcg@10'] - float(ref['test_ndcg@10'])) < 0.02 and abs(res['recall@20'] - float(ref['test_recall@20'])) < 0.03
    got = np.array([h['train_loss'] for h in hist[:12]])
    np.testing.assert_allclose(got, ref['train_loss'][:12], rtol=0.02)
    got_v = np.array([h['ndcg@5'] for h in hist[:12]])
    assert np.abs(got_v[3:] - ref['val_ndcg@5'][3:12]).max() < 0.03


@pytest.mark.parametrize('kind', ['adam', 'sgd'])
def test_bpr_fit_ml100k_fused_optimizers_reach_the_same_band(ra, kind):
    """The in-kernel optimizers on the same data (``train.fused_optimizer``; they need whole 64-negative tiles, so n = 64
    instead of the stock 1): lazy Adam (SparseAdam's rule) and plain in-place SGD must converge to the band the
    reference's dense Adam reaches -- a row-sparse update rule that silently lost updates would not."""
    train = {'negative_count': 64, 'fused_optimizer': kind}
    if kind == 'sgd':
        train.update(learning_rate=20.0, epochs=150)
    model = ra.BPR({'eval': {'batch_size': 20}, 'train': train})
    ds = _ml100k(ra)
    trn, val, tst = ds.build(split_ratio=[0.8, 0.1, 0.1], shuffle=True, split_mode='user_entry')
    model.fit(trn, val)
    res = model.evaluate(tst, verbose=False)
    print(kind, 'epochs', len(model.history), 'test', {k: round(float(v), 4) for k, v in res.items() if k.endswith('@10') or k.endswith('@20')})
    assert 0.21 <= res['ndcg@10'] <= 0.29 and 0.31 <= res['recall@20'] <= 0.41


# --------------------------------------------------------------------------- item-tower models under the sharded fit, HIP backend
def _sasrec_gpu_worker(rank, world, port, result_dir, layout):
    import torch.distributed as dist
    import recstudio_amd as ra
    from recstudio_amd import shard
    from staged_dist import StagedDist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        # shard_rows_share = 1: the tower's fixed-capacity look-up segments hold a whole call (this test compares trajectories to
        # 5e-5: a gated overflow step under a torch optimizer would show; the overflow path itself: tests/test_shard_gloo.py)
        conf = {'train': {'epochs': 2, 'batch_size': 2048 // world, 'seed': 2022, 'learning_rate': 0.003, 'early_stop_patience': 100,
                          'shard_layout': layout, 'shard_rows_share': 1.0},
                'eval': {'batch_size': 128 // world, 'cutoff': [10], 'val_metrics': ['ndcg', 'recall'], 'topk': 50,
                         'test_metrics': ['ndcg', 'recall']},
                'model': {'embed_dim': 64, 'dropout_rate': 0.0}}
        model = ra.SASRec(conf)                          # stock: BinaryCrossEntropyLoss (autograd over the exchanged scores), n = 1
        sq = _ml100k(ra, ra.SeqDataset, max_seq_len=20)
        trn, val, tst = sq.build(split_ratio=2)
        trn.data_index = trn.data_index[:len(trn.data_index) // 2 * 2]      # no padded (repeated-sample) last batch
        seen = []
        orig = shard.allreduce_grads

        def spy(params, *a, **k):
            params = list(params)
            seen.extend(tuple(p.shape) for p in params)
            return orig(params, *a, **k)
        shard.allreduce_grads = spy
        torch.cuda.manual_seed_all(2022)
        best = model.fit(trn, val, dist=StagedDist(dist), device='cuda:0')
        test = model.evaluate(tst, verbose=False)
        assert isinstance(model.query_encoder.item_encoder, shard.ShardedRows)
        assert all(s[0] != trn.num_items for s in seen)                      # no [N, d] tensor was all-reduced
        assert tuple(model.item_encoder.weight.shape) == (model._shard['plan'].n_local(rank), 64)
        dense = torch.cat([p.detach().reshape(-1) for p in model.query_encoder.parameters()]).cpu()
        torch.save({'best': best, 'val': dict(model.logged_metrics), 'test': test, 'losses': torch.cat(model.train_losses),
                    'item': model.item_encoder.weight.detach().cpu(), 'tower': dense}, os.path.join(result_dir, f'w{world}r{rank}.pt'))
    finally:
        dist.destroy_process_group()


def test_sasrec_fit_two_staged_ranks_equals_one_rank_hip(tmp_path):
    """VERDICT r3 #1 with the HIP backend: stock SASRec (history embedded with the scored table, BCE loss under autograd
    over the exchanged scores) through ``BaseRetriever.fit`` / ``evaluate`` as two ranks sharing the test GPU (collectives
    staged over gloo) == the one-rank run: losses, metrics, Transformer weights and the tied item table; the tower's
    look-ups go through ``rsa_embedding_gather`` on the owner and the sorted row scatter in the backward."""
    import torch.multiprocessing as mp
    for world, layout in ((1, 'block'), (2, 'block'), (2, 'interleaved')):
        sub = tmp_path / f'{layout}{world}'
        os.makedirs(sub)
        mp.spawn(_sasrec_gpu_worker, args=(world, _free_port(), str(sub), layout), nprocs=world, join=True)
    one = torch.load(tmp_path / 'block1' / 'w1r0.pt', weights_only=False)
    assert float(one['losses'][-1]) < float(one['losses'][0]) - 0.05 and one['val']['ndcg@10'] > 0.005
    for layout in ('block', 'interleaved'):
        two = [torch.load(tmp_path / f'{layout}2' / f'w2r{r}.pt', weights_only=False) for r in range(2)]
        for t in two:
            np.testing.assert_allclose(t['losses'].numpy(), one['losses'].numpy(), rtol=5e-5, atol=1e-6)
            for k in ('ndcg@10', 'recall@10'):
                assert abs(t['val'][k] - one['val'][k]) < 2e-3 and abs(t['test'][k] - one['test'][k]) < 2e-3
            # (Adam at 3e-3: a weight whose gradient is near zero moves by up to the rate on a summation-order difference)
            np.testing.assert_allclose(t['tower'].numpy(), one['tower'].numpy(), rtol=5e-3, atol=1e-3)
        items = torch.empty_like(one['item'])
        if layout == 'interleaved':
            items[0::2], items[1::2] = two[0]['item'], two[1]['item']
        else:
            items = torch.cat([two[0]['item'], two[1]['item']])
        np.testing.assert_allclose(items.numpy(), one['item'].numpy(), rtol=5e-3, atol=1e-3)
        assert not items[0].any()


def test_sharded_topk_wide_history_hip(ra):
    """``_topk_sharded`` with k + |history| beyond the in-kernel select (1024): the 512-candidate pass, and -- for
    a user whose history IS the head of the ranking -- the wide pass (materialised scores, stable-sort merge), at world 1
    through the HIP backend; == oracle."""
    import torch.distributed as dist
    from recstudio_amd.shard import RowShardPlan, ShardedItemTable
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        n_items, d, B, k, width = 5001, 64, 6, 10, 1100
        g = torch.Generator().manual_seed(9)
        item = torch.randn(n_items, d, generator=g)
        item[0] = 0
        q = torch.randn(B, d, generator=g)
        table = ShardedItemTable(item.to(DEV), RowShardPlan(n_items, 1), 0, dist)
        model = ra.BaseRetriever({'train': {'seed': None}})
        model._shard = {'table': table, 'n_items': n_items, 'hist_width': width}
        best = torch.argsort(-(q @ item[1:].t()), dim=1) + 1
        for blocked in (False, True):
            hist = torch.zeros(B, width, dtype=torch.int64)
            hist[:, :300] = best[:, 600:900]
            if blocked:
                hist[1, :1000] = best[1, :1000]            # the whole head of the ranking is history: 512 candidates run dry
            score, ids = model._topk_sharded(q.to(DEV), k, hist.to(DEV), False)
            want_s, want_i = oracle.topk_with_history(q, item, k, hist)
            assert torch.equal(ids.cpu(), want_i)
            rel_close(score.cpu(), want_s, rtol=1e-4, atol=1e-5)
    finally:
        dist.destroy_process_group()


# --------------------------------------------------------------------------- the owner side of the sharded backward, one call
@pytest.mark.parametrize('d,n_rows,Q,n_seg,cap', [(128, 5003, 96, 8, 700), (64, 301, 7, 3, 90), (256, 100_003, 2048, 16, 9000),
                                                  (128, 17, 4, 2, 3000), (128, 50_021, 64, 8, 20_000)])
def test_owner_backward_segments_vs_index_add(ra, d, n_rows, Q, n_seg, cap):
    """rsa_shard_backward_segments (in-tree radix sorts by row and by query straight from the segments, one walk over the
    query runs, solo rows in place, sorted apply for the shared rows) == two index_add_ in float64: the dense gradient
    block, plain SGD in place, the padding row, dead slack (NaN coefficients there: never read), a gated step, and
    bit-equality run to run.  Shapes: few long runs (a 17-row block: every row shared by hundreds of slots), many short
    runs, runs longer than a tile per wave (cap 20 000 over 64 queries)."""
    from recstudio_amd.shard import HipBackend
    be = HipBackend()
    g = torch.Generator().manual_seed(d + n_rows)
    stride = cap + be.HDR
    keys = torch.full((n_seg, stride), -7, dtype=torch.int64)
    dv = torch.full((n_seg, stride), float('nan'))
    live = torch.randint(0, cap + 1, (n_seg,), generator=g)
    live[0] = cap                                                          # a full segment ...
    if n_seg > 1:
        live[1] = 0                                                        # ... and an empty one
    rows_l, q_l, d_l = [], [], []
    for s_ in range(n_seg):
        c = int(live[s_])
        keys[s_, 0], keys[s_, 1] = c, 0
        r = torch.randint(0, n_rows, (c,), generator=g)
        if c > 10:
            r[:5] = 0                                                      # the padding row gets slots too
        q = torch.randint(0, Q, (c,), generator=g)
        keys[s_, be.HDR:be.HDR + c] = (q << 32) | r
        dd = torch.randn(c, generator=g)
        dv[s_, be.HDR:be.HDR + c] = dd
        rows_l.append(r), q_l.append(q), d_l.append(dd)
    rows, qi, dd = torch.cat(rows_l), torch.cat(q_l), torch.cat(d_l)
    item = torch.randn(n_rows, d, generator=g) * 0.3
    q_all = torch.randn(Q, d, generator=g) * 0.3
    keep = rows != 0
    want_q = torch.zeros(Q, d, dtype=torch.float64).index_add_(0, qi, dd.double().unsqueeze(1) * item[rows].double())
    want_i = torch.zeros(n_rows, d, dtype=torch.float64).index_add_(0, rows[keep], dd[keep].double().unsqueeze(1) * q_all[qi[keep]].double())
    keys_d, dv_d, q_d = keys.to(DEV).view(-1), dv.to(DEV).view(-1), q_all.to(DEV)

    def run(target_is_table, lr=None, dropped=0):
        st = be.new_state(DEV)
        st['step_dropped'].fill_(dropped)
        tab = item.to(DEV).clone()
        tgt = tab if target_is_table else torch.zeros(n_rows, d, device=DEV)
        qg = torch.zeros(Q, d, device=DEV)
        sc = None if lr is None else torch.full((1,), -lr, device=DEV)
        be.backward_segments(st, tab, q_d, keys_d, n_seg, stride, dv_d, tgt, qg, item_pad_row=0, item_scale=sc)
        return tab, tgt, qg, st
    # dense gradient block
    tab, grad, qg, st = run(False)
    assert torch.equal(tab.cpu(), item)                                    # the table is only read
    rel_close(qg.cpu(), want_q.float(), rtol=2e-4, atol=2e-5)
    rel_close(grad.cpu(), want_i.float(), rtol=2e-4, atol=2e-5)
    assert not grad[0].any() and st['scale'].tolist() == [1.0, 1.0]
    _, grad2, qg2, _ = run(False)
    assert torch.equal(grad, grad2) and torch.equal(qg, qg2)                # deterministic
    # SGD in place: solo rows by the walk, shared rows by the sorted apply, the padding row untouched
    tab, _, qg3, st = run(True, lr=0.25)
    rel_close(tab.cpu(), (item.double() - 0.25 * want_i).float(), rtol=2e-4, atol=2e-5)
    rel_close(qg3.cpu(), want_q.float(), rtol=2e-4, atol=2e-5)
    assert torch.equal(tab[0].cpu(), item[0]) and st['scale'].tolist() == [-0.25, 1.0]
    tab_b, _, qg3b, _ = run(True, lr=0.25)
    assert torch.equal(tab, tab_b) and torch.equal(qg3, qg3b)
    # a step in which some rank dropped an element changes nothing
    tab, _, qg4, st = run(True, lr=0.25, dropped=3)
    assert torch.equal(tab.cpu(), item) and not qg4.any() and st['scale'].tolist() == [0.0, 0.0]


@pytest.mark.parametrize('n_items,total_q,n', [(97, 41, 64), (2 ** 24 + 5, 300, 64), (100_000_001, 64, 128), (1000, 70_000, 64)])
def test_in_tree_radix_sort_through_the_sorted_scatter(ra, n_items, total_q, n):
    """The in-tree LSD radix sort (rsa_radix.hpp; 1 .. 4 passes of 8 bits: 97 items -> 1 pass, 2^24 + 5 -> 4, 1e8 -> 4) under
    the sorted scatter: every element lands on its row (== index_add_ in float64 on the touched rows), ids beyond 2^24,
    negative ids (empty slots) dropped, a padding row skipped, more than one tile per digit (70 000 x 65 elements)."""
    d = 64
    g = torch.Generator().manual_seed(n_items % 1000)
    ids = torch.randint(0, n_items, (total_q, n), generator=g)
    ids[:, :3] = torch.tensor([n_items - 1, 0, n_items // 2])
    ids[::7, 5] = -1                                                       # empty slots
    pos = torch.randint(1, n_items, (total_q,), generator=g)
    dneg, dpos = torch.randn(total_q, n, generator=g), torch.randn(total_q, generator=g)
    q = torch.randn(total_q, d, generator=g)
    # a compact table: the touched rows only (a 1e8 x 64 target would be 25 GB; the sort does not care about the table)
    touched = torch.unique(torch.cat([ids.reshape(-1).clamp(min=0), pos]))
    remap = {int(v): i for i, v in enumerate(touched.tolist())}
    if n_items <= 2 ** 24 + 5 and n_items * d * 4 < 6e9:
        target = torch.zeros(n_items, d, device=DEV)
        ra.ops.scatter_rows_sorted(target, q.to(DEV), ids.to(DEV), dneg.to(DEV), pos_ids=pos.to(DEV), dpos=dpos.to(DEV), pad_row=0)
        got = target[touched.to(DEV)].cpu()
    else:
        # the full-size key range through the sort + classification entry point: solo flags == a CPU count
        solo = ra.ops.sort_step_elements(pos.to(DEV), ids.to(DEV), n_items, pad_row=0)[0].cpu().view(total_q, n + 1)
        flat = torch.cat([pos.view(-1, 1), ids], 1)
        uniq, cnt = torch.unique(flat[flat >= 0], return_counts=True)
        once = set(uniq[cnt == 1].tolist()) - {0}
        want = torch.tensor([[int(v) in once for v in row] for row in flat.tolist()])
        assert torch.equal(solo.bool(), want)
        return
    want = torch.zeros(len(remap), d, dtype=torch.float64)
    live = ids.reshape(-1) > 0
    rows = torch.tensor([remap[int(v)] for v in ids.reshape(-1)[live].tolist()])
    want.index_add_(0, rows, dneg.reshape(-1)[live].double().unsqueeze(1) * q.repeat_interleave(n, 0)[live].double())
    want.index_add_(0, torch.tensor([remap[int(v)] for v in pos.tolist()]), dpos.double().unsqueeze(1) * q.double())
    # (a 1000-row catalog under 4.5 M elements: every row sums ~4500 terms in fp32, the tolerance follows the magnitude)
    rel_close(got, want.float(), rtol=2e-4, atol=2e-6 * float(want.abs().max()) + 2e-5)


# --------------------------------------------------------------------------- fit on a big catalog: no [N, d] on the host
def _big_catalog_worker(rank, world, port, result_dir, mode):
    import torch.distributed as dist
    import recstudio_amd as ra
    from staged_dist import StagedDist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        n_items, n_users, d = 3_000_000, 5_000, 128
        g = np.random.default_rng(5)
        items = np.concatenate([np.arange(1, n_items + 1), g.integers(1, n_items + 1, size=200_000)])   # every item seen once
        users = g.integers(1, n_users + 1, size=items.size)
        model = ra.BPR({'model': {'embed_dim': d}, 'train': {'epochs': 1, 'batch_size': 65_536 // world, 'negative_count': 64,
                                                            'shard_init': mode, 'learning_rate': 0.0},      # (rate 0: the weights stay the initial ones)
                        'eval': {'batch_size': 64, 'topk': 20, 'cutoff': [10]}})
        ds = ra.TripletDataset.from_interactions(users, items)
        trn, val, _ = ds.build(split_ratio=[0.98, 0.01, 0.01], shuffle=False)
        assert trn.num_items == n_items + 1
        # (first use of the GPU maps the runtime's and the library's code objects into the process: before the baseline)
        warm = torch.randn(64, d, device='cuda:0')
        ra.ops.embedding_gather(warm, torch.arange(8, device='cuda:0'))
        torch.nn.Embedding(8, d).to('cuda:0')(torch.arange(8, device='cuda:0')).sum().item()
        torch.cuda.synchronize()
        # the resident set while fit() runs, sampled every 2 ms (the process's all-time peak was set by the dataset build)
        import threading
        import psutil
        proc, stop, seen = psutil.Process(), threading.Event(), [0]

        def watch():
            while not stop.is_set():
                seen[0] = max(seen[0], proc.memory_info().rss)
                stop.wait(0.002)
        before = proc.memory_info().rss
        th = threading.Thread(target=watch, daemon=True)
        th.start()
        model.fit(trn, None, dist=StagedDist(dist), device='cuda:0')
        stop.set()
        th.join()
        w = model.item_encoder.weight.detach()
        torch.save({'grew_MiB': (seen[0] - before) / 2 ** 20, 'rows': w[:1000].cpu(), 'std': float(w.std()),
                    'loss': float(model.logged_metrics['train_loss'])}, os.path.join(result_dir, f'{mode}_w{world}r{rank}.pt'))
    finally:
        dist.destroy_process_group()


def test_fit_on_a_big_catalog_never_holds_the_table_on_the_host(tmp_path):
    """VERDICT r3 missing #5: ``BaseRetriever.fit`` as one rank of a job over a catalog whose table is large (here 3 M x
    128 fp32 = 1.5 GB; the default switches at 2 GiB): with ``train.shard_init: 'device'`` the host's peak resident set grows
    by far less than one copy of the table during ``fit`` (the host-initialised form holds the table plus this rank's
    slice), two ranks hold the rows of the table one rank draws, and the model trains."""
    import torch.multiprocessing as mp
    for mode, world in (('device', 1), ('device', 2), ('host', 1)):
        mp.spawn(_big_catalog_worker, args=(world, _free_port(), str(tmp_path), mode), nprocs=world, join=True)
    dev1 = torch.load(tmp_path / 'device_w1r0.pt', weights_only=False)
    dev2 = torch.load(tmp_path / 'device_w2r0.pt', weights_only=False)
    host = torch.load(tmp_path / 'host_w1r0.pt', weights_only=False)
    table_MiB = 3_000_001 * 128 * 4 / 2 ** 20
    print('host RSS growth during fit, MiB:', {'device': dev1['grew_MiB'], 'device, 2 ranks': dev2['grew_MiB'], 'host': host['grew_MiB']})
    # measured on the test box: host-initialised 3114 MiB (the table + this rank's slice of it), device-initialised 506 MiB
    # with one rank, 861 MiB with two (the staged test collectives copy through host memory; RCCL does not)
    assert host['grew_MiB'] > 1.5 * table_MiB                                 # the full table lived on the host, twice
    assert dev1['grew_MiB'] < 0.5 * table_MiB and dev2['grew_MiB'] < 0.75 * table_MiB
    assert abs(dev1['std'] - (2.0 / (3_000_001 + 128)) ** 0.5) < 2e-4          # xavier_normal over the full shape (init.py)
    assert np.isfinite(dev1['loss']) and np.isfinite(dev2['loss']) and abs(dev1['loss'] - dev2['loss']) < 1e-3
    # (learning rate 0) two ranks drew the rows of the table one rank draws, bit for bit
    assert torch.equal(dev1['rows'], dev2['rows']) and not dev1['rows'][0].any() and float(dev1['rows'][1:].abs().min()) >= 0


@pytest.mark.parametrize('n,B,layout', [(64, 512, 'block'), (256, 96, 'block'), (1024, 40, 'interleaved'), (128, 320, 'block'),
                                        (128, 300, 'block')])
def test_query_grouped_routing_needs_no_sort_by_query(ra, n, B, layout):
    """`rsa_shard_route_args.group_by_query`: when n divides 1024 a routing workgroup holds whole queries and writes its
    share of every segment query by query, so the owner reads the queries' runs off the slots (`keys_grouped`) instead of
    sorting them -- the training step (loss share, in-place-updated block, user rows, drawn ids) equals the sorted form's,
    popularity and uniform sampler.  (128, 300): the draw's grid is not a multiple of 1024 threads -- the router says so
    (`rsa_shard_route_query_groups` = 0) and the step falls back to the sort."""
    import torch.distributed as dist
    from recstudio_amd.shard import RowShardPlan, ShardedItemTable, ShardedRetriever
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        N, U, d = 30_011, 500, 128
        g = torch.Generator().manual_seed(n + B)
        item0 = (torch.randn(N, d, generator=g) * 0.2)
        item0[0] = 0
        user0 = torch.randn(U, d, generator=g) * 0.2
        uid = torch.randint(1, U, (B,), generator=g).to(DEV)
        pos = torch.randint(1, N, (B,), generator=g).to(DEV)
        counts = (torch.rand(N, generator=g) ** 3 * 50).long()
        for sampler in (ra.UniformSampler(N), ra.PopularSamplerModel(counts).to(DEV)):
            res = {}
            for grouped in (True, False):
                item = item0.to(DEV).clone()
                tower = torch.nn.Embedding(U, d).to(DEV)
                with torch.no_grad():
                    tower.weight.copy_(user0)
                table = ShardedItemTable(item, RowShardPlan(N, 1, layout=layout), 0, dist, check_every=0, sample_seed=5)
                table.group_by_query = grouped
                seen = []
                fwd = table.backend.owner_bpr_forward
                table.backend.owner_bpr_forward = lambda *a, **k: (seen.append(k.get('keys_grouped')), fwd(*a, **k))[1]
                trainer = ShardedRetriever(table, tower, sampler, ra.BPRLoss(), n, item_sgd_lr=0.3, query_sgd_lr=0.3, keep_neg_ids=True)
                losses = [float(trainer.training_step(uid, pos)) for _ in range(2)]
                can_group = (B * n) % 1024 == 0
                assert seen == [grouped and can_group] * 2
                res[grouped] = (losses, item.clone(), tower.weight.detach().clone(), trainer.last_neg.clone())
            a, b = res[True], res[False]
            assert torch.equal(a[3], b[3])                                  # the same draw
            rel_close(a[0], b[0], rtol=1e-6)
            rel_close(a[1].cpu(), b[1].cpu(), rtol=1e-5, atol=1e-7)
            rel_close(a[2].cpu(), b[2].cpu(), rtol=1e-5, atol=1e-7)
            assert (a[1].cpu() - item0).abs().max() > 1e-4                  # the step really trained
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('d,n,B,layout,popular', [(64, 64, 300, 'block', False), (128, 256, 64, 'interleaved', True), (256, 128, 96, 'block', False),
                                                  (128, 1024, 24, 'block', True), (128, 100, 77, 'block', False), (64, 512, 40, 'interleaved', False)])
def test_bpr_step_on_owners_vs_autograd(ra, d, n, B, layout, popular):
    """The BPR training step evaluated on the owner (world size 1; every tile width 16 / 32 / 64 lanes per row, 1 / 2 / 4
    waves per query run, grouped routing and the sort fallback (n = 100), both samplers, both row layouts): loss, the dense
    item-gradient block and the user-row gradient == torch autograd of loss_func.py:55-59 on the same negatives; then the
    same step with SGD in place == weights minus rate times that gradient."""
    import torch.distributed as dist
    from recstudio_amd.shard import RowShardPlan, ShardedItemTable, ShardedRetriever
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        N, U = 20_011, 300
        g = torch.Generator().manual_seed(d + n)
        item0 = torch.randn(N, d, generator=g) * 0.2
        item0[0] = 0
        user0 = torch.randn(U, d, generator=g) * 0.2
        uid = torch.randint(1, U, (B,), generator=g).to(DEV)
        pos = torch.randint(1, N, (B,), generator=g).to(DEV)
        counts = (torch.rand(N, generator=g) ** 3 * 50).long()
        sampler = ra.PopularSamplerModel(counts).to(DEV) if popular else ra.UniformSampler(N)

        def make(**kw):
            item = item0.to(DEV).clone()
            tower = torch.nn.Embedding(U, d).to(DEV)
            with torch.no_grad():
                tower.weight.copy_(user0)
            table = ShardedItemTable(item, RowShardPlan(N, 1, layout=layout), 0, dist, check_every=0, sample_seed=9)
            return item, tower, ShardedRetriever(table, tower, sampler, ra.BPRLoss(), n, sparse_query_rows=True, keep_neg_ids=True, **kw)
        item, tower, tr = make()
        assert tr.table.owner_loss_ok()
        loss = tr.training_step(uid, pos)
        neg = tr.last_neg
        iw = item0.to(DEV).clone().requires_grad_(True)
        uw = user0.to(DEV).clone().requires_grad_(True)
        q = uw[uid]
        ref = -torch.nn.functional.logsigmoid((q * iw[pos]).sum(-1, keepdim=True) - (q.unsqueeze(1) * iw[neg]).sum(-1)).mean(-1).mean()
        ref.backward()
        rel_close(loss.item(), ref.item(), rtol=1e-5)
        want = iw.grad.clone()
        want[0] = 0
        rel_close(tr.item_grad_local.cpu(), want.cpu(), rtol=2e-4, atol=1e-7)
        rel_close(tr.query_grad_dense().cpu(), uw.grad.cpu(), rtol=2e-4, atol=1e-7)
        assert torch.equal(item.cpu(), item0)                              # the gradient-block form only reads the table
        item2, tower2, tr2 = make(item_sgd_lr=0.5, query_sgd_lr=0.5)
        loss2 = tr2.training_step(uid, pos)
        # (same draw; the loss to rounding: the order of a query's elements inside its run follows the router's LDS atomics)
        assert torch.equal(tr2.last_neg, neg)
        rel_close(loss2.item(), loss.item(), rtol=1e-6)
        rel_close(item2.cpu(), (item0.to(DEV) - 0.5 * want).cpu(), rtol=2e-4, atol=1e-7)
        rel_close(tower2.weight.detach().cpu(), (user0.to(DEV) - 0.5 * uw.grad).cpu(), rtol=2e-4, atol=1e-7)
        # the query rows' update on the second stream (default) and behind the apply pass: the same step
        item3, tower3, tr3 = make(item_sgd_lr=0.5, query_sgd_lr=0.5, overlap_query_rows=False)
        tr3.training_step(uid, pos)
        assert tr2._side is not None and tr3._side is None
        rel_close(item3.cpu(), item2.cpu(), rtol=1e-5, atol=1e-7)
        rel_close(tower3.weight.detach().cpu(), tower2.weight.detach().cpu(), rtol=1e-5, atol=1e-7)
    finally:
        dist.destroy_process_group()


# --------------------------------------------------------------------------- SASRec tower against the reference's own module
def test_sasrec_tower_and_tied_gradients_vs_reference(ra):
    """SURVEY 8a E3 / configs[2] pinned by the REAL reference (tests/golden/sasrec.npz, oracle/make_golden_sasrec.py:
    recstudio/model/seq/sasrec.py:8-67 + InnerProductScorer + SampledSoftmaxLoss / BinaryCrossEntropyLoss on a fixed batch
    with fixed weights): the tower's output (history rows through the HIP gather, stock Transformer), the scores of given
    negatives, both losses, and the gradients -- of the TIED item table (history gather + positives + negatives, through the
    sorted scatters), of the position table and of a Transformer weight -- in training and in evaluation mode."""
    from recstudio_amd.retriever import SASRecQueryEncoder
    g = np.load(os.path.join(HERE, 'golden', 'sasrec.npz'))
    N, d, L = int(g['N']), int(g['d']), int(g['L'])
    item = torch.nn.Embedding(N, d, padding_idx=0)
    enc = SASRecQueryEncoder('item_id', d, L, int(g['heads']), int(g['hidden']), 0.0, 'gelu', 1e-12, int(g['layers']), item)
    state = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('w::')}
    assert set(state) == set(enc.state_dict())                  # the reference module's parameter names, one for one
    enc.load_state_dict(state)
    enc.to(DEV).train()
    hist, seqlen = torch.from_numpy(g['hist']).to(DEV), torch.from_numpy(g['seqlen']).to(DEV)
    pos, neg = torch.from_numpy(g['pos']).to(DEV), torch.from_numpy(g['neg']).to(DEV)
    log_pos, log_neg = torch.from_numpy(g['log_pos']).to(DEV), torch.from_numpy(g['log_neg']).to(DEV)
    batch = {'in_item_id': hist, 'seqlen': seqlen}
    n = neg.shape[1]
    # Two comparisons.  (i) the reference fixture (recorded on the CPU): semantics -- masking, pooling, the tie -- at the
    # tolerance the stock PyTorch-ROCm Transformer itself meets against the CPU (its fp32 GEMMs and attention differ from
    # the CPU's in the 4th digit after two layers; out of scope).  (ii) the same op sequence with stock torch
    # ops ON THIS DEVICE (F.embedding for the history, torch's scorer arithmetic and autograd): isolates this package's
    # kernels -- the gather is exact, scores / loss 1e-5, the tied table's gradient 2e-4.
    import copy
    ref_item = copy.deepcopy(item)
    ref_enc = copy.deepcopy(enc)
    ref_enc.item_encoder = ref_item                              # (re-tie the copy)

    def stock_tower(batch):
        hist_ = batch['in_item_id']
        Ln = hist_.shape[1]
        seq = torch.nn.functional.embedding(hist_, ref_item.weight, padding_idx=0) + ref_enc.position_emb.weight[:Ln].unsqueeze(0)
        causal = torch.triu(torch.ones(Ln, Ln, dtype=torch.bool, device=hist_.device), 1)
        o = ref_enc.transformer_layer(ref_enc.dropout(seq), mask=causal, src_key_padding_mask=hist_ == 0)
        last = (batch['seqlen'] - 1).view(-1, 1, 1).expand(-1, 1, o.shape[-1])
        return o.gather(1, last).squeeze(1)
    for tag, loss_fn in (('ssm', ra.SampledSoftmaxLoss()), ('bce', ra.BinaryCrossEntropyLoss())):
        enc.zero_grad()
        query = enc(batch)
        score, _ = ra.retriever_scores(item.weight, query, n, pos_ids=pos, neg_ids=neg)
        loss = loss_fn(None, score['pos_score'], log_pos, score['neg_score'], log_neg)
        loss.backward()
        # (i) the reference's recorded run
        rel_close(query.detach().cpu(), g[tag + '_query'], rtol=3e-3, atol=3e-4)
        rel_close(score['neg_score'].detach().cpu(), g[tag + '_neg_score'], rtol=3e-3, atol=3e-4)
        rel_close(loss.item(), float(g[tag + '_loss']), rtol=1e-3)
        rel_close(item.weight.grad.cpu(), g[tag + '_item_grad'], rtol=2e-2, atol=3e-5)
        rel_close(enc.position_emb.weight.grad.cpu(), g[tag + '_posemb_grad'], rtol=2e-2, atol=3e-5)
        assert not item.weight.grad[0].any()                    # the padding row never receives gradient
        # (ii) stock torch ops on this device
        ref_enc.zero_grad()
        ref_item.zero_grad()
        q2 = stock_tower(batch)
        ps2 = (q2 * ref_item(pos)).sum(-1)
        ns2 = (q2.unsqueeze(1) * ref_item(neg)).sum(-1)
        l2 = (oracle.sampled_softmax_loss(ps2, log_pos, ns2, log_neg) if tag == 'ssm' else oracle.bce_loss(ps2, ns2))
        l2.backward()
        rel_close(query.detach().cpu(), q2.detach().cpu(), rtol=1e-5, atol=1e-6)
        rel_close(score['neg_score'].detach().cpu(), ns2.detach().cpu(), rtol=1e-4, atol=1e-6)
        rel_close(loss.item(), l2.item(), rtol=1e-5)
        rel_close(item.weight.grad.cpu(), ref_item.weight.grad.cpu(), rtol=2e-4, atol=1e-7)
        rel_close(enc.position_emb.weight.grad.cpu(), ref_enc.position_emb.weight.grad.cpu(), rtol=2e-4, atol=1e-7)
        rel_close(enc.transformer_layer.layers[0].linear1.weight.grad.cpu(),
                  ref_enc.transformer_layer.layers[0].linear1.weight.grad.cpu(), rtol=2e-4, atol=1e-7)
    enc.eval()
    with torch.no_grad():
        rel_close(enc(batch).cpu(), g['eval_query'], rtol=3e-3, atol=3e-4)
