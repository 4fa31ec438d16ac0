"""GPU: full-catalog scoring (fp32 MFMA), fused logsumexp, exact top-k and history masking
against the oracle and the reference fixtures (tests/golden/topk.npz, forward.npz)."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = 'cuda'


@pytest.fixture(scope='module')
def ra():
    import recstudio_amd
    recstudio_amd._native.lib()
    torch.cuda.init()
    return recstudio_amd


def close(a, b, rtol=1e-4, atol=1e-5):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


@pytest.mark.parametrize('d', [32, 64, 128, 16, 100])
@pytest.mark.parametrize('N,B', [(3, 1), (34, 3), (1000, 37), (5003, 200), (40000, 129)])
def test_scores_lse_topk_vs_oracle(ra, d, N, B):
    g = torch.Generator().manual_seed(N + B + d)
    iw = torch.randn(N, d, generator=g) * 0.5
    iw[0] = 7.0            # the padding row must never be scored
    q = torch.randn(B, d, generator=g) * 0.5
    want = oracle.inner_product_score(q, iw[1:])            # [B, N-1]
    k = min(100, N - 1)
    scores, lse, tv, ti = ra.ops.fullscore(iw.to(DEV), q.to(DEV), want_scores=True, want_lse=True, k=k)
    close(scores.cpu(), want)
    close(lse.cpu(), torch.logsumexp(want, -1), rtol=1e-5)
    wv, wi = torch.topk(want, k)
    close(tv.cpu(), wv)
    got_i = ti.cpu()
    # same items wherever the neighbouring scores are separated by more than the fp32 noise
    gap_ok = torch.ones_like(wv, dtype=torch.bool)
    if k > 1:
        gaps = (wv[:, :-1] - wv[:, 1:]).abs() > 1e-5
        gap_ok[:, 1:] &= gaps
        gap_ok[:, :-1] &= gaps
    assert torch.equal(got_i[gap_ok], (wi + 1)[gap_ok])
    # fused outputs without materialising the score matrix give the same answers
    _, lse2, tv2, ti2 = ra.ops.fullscore(iw.to(DEV), q.to(DEV), want_lse=True, k=k)
    assert torch.equal(lse2, lse) and torch.equal(tv2, tv) and torch.equal(ti2, ti)
    # item_vector view (weight[1:]) without a copy
    s3 = ra.ops.fullscore(iw.to(DEV)[1:], q.to(DEV), want_scores=True, items_without_pad=True)[0]
    assert torch.equal(s3, scores)


def test_topk_ties_and_large_k(ra):
    N, d, B = 3000, 32, 5
    iw = torch.zeros(N, d)
    iw[1:, 0] = (torch.arange(N - 1) % 7).float()          # many exact ties
    q = torch.zeros(B, d)
    q[:, 0] = 1.0
    k = 1000
    _, _, tv, ti = ra.ops.fullscore(iw.to(DEV), q.to(DEV), k=k)
    want = oracle.inner_product_score(q, iw[1:])
    wv, _ = torch.topk(want, k)
    assert torch.equal(tv.cpu(), wv)
    # ties resolve to the smaller item id first, and returned ids really carry the returned score
    assert torch.equal(want.gather(1, ti.cpu() - 1), tv.cpu())
    for b in range(B):
        ids = ti[b].cpu()
        vals = tv[b].cpu()
        for v in vals.unique():
            grp = ids[vals == v]
            assert torch.equal(grp, grp.sort().values)


def test_topk_with_history_golden(ra, golden):
    """BaseRetriever.topk fixture recorded from the reference (history masking included)."""
    g = golden('topk')
    iw, uw, uid, hist = (T(g[k]).to(DEV) for k in ('item_w', 'user_w', 'uid', 'hist'))
    k = g['items'].shape[1]
    q = ra.ops.embedding_gather(uw, uid)
    more = hist.shape[1]
    _, _, cv, ci = ra.ops.fullscore(iw, q, k=k + more)
    val, idx = ra.ops.topk_mask_history(cv, ci, hist, k)
    assert np.array_equal(idx.cpu().numpy(), g['items'])
    close(val.cpu(), g['score'], rtol=1e-5)
    _, _, cv, ci = ra.ops.fullscore(iw, q, k=k)
    assert np.array_equal(ci.cpu().numpy(), g['items_nohist'])
    close(cv.cpu(), g['score_nohist'], rtol=1e-5)


def test_mask_history_fewer_than_k_survivors(ra):
    cv = torch.tensor([[5., 4., 3., 2., 1.]])
    ci = torch.tensor([[10, 11, 12, 13, 14]])
    hist = torch.tensor([[11, 13, 14, 0]])
    v, i = ra.ops.topk_mask_history(cv.to(DEV), ci.to(DEV), hist.to(DEV), 4)
    assert v[0, :2].tolist() == [5., 3.] and i[0, :2].tolist() == [10, 12]
    assert torch.isinf(v[0, 2:]).all() and sorted(i[0, 2:].tolist()) in ([11, 13], [11, 14], [13, 14])


def test_softmax_loss_full_catalog_golden(ra, golden):
    g = golden('forward')
    tag = 'softmax_ip'
    iw = T(g[tag + '_item_w']).to(DEV).requires_grad_(True)
    uw = T(g[tag + '_user_w']).to(DEV).requires_grad_(True)
    uid, pos = T(g[tag + '_uid']).to(DEV), T(g[tag + '_pos']).to(DEV)
    q = torch.nn.functional.embedding(uid, uw)               # stock torch tower, as in the reference
    all_score = ra.InnerProductScorer()(q, iw[1:])
    close(all_score.detach().cpu(), g[tag + '_all_score'], atol=1e-6)
    pos_score = (q * iw[pos]).sum(-1)
    loss = ra.SoftmaxLoss()(label=None, pos_score=pos_score, all_score=all_score)
    close(loss.detach().cpu(), g[tag + '_loss'], rtol=1e-5)
    loss.backward()
    close(iw.grad.cpu(), g[tag + '_item_grad'], rtol=1e-4, atol=1e-7)
    close(uw.grad.cpu(), g[tag + '_user_grad'], rtol=1e-4, atol=1e-7)


def test_full_softmax_without_score_matrix_golden(ra, golden):
    """full_lse (forward never writes [B, N]) + BaseRetriever.training_step dispatch == the reference's
    SoftmaxLoss value and dense gradients."""
    g = golden('forward')
    tag = 'softmax_ip'
    iw = T(g[tag + '_item_w']).to(DEV).requires_grad_(True)
    uw = T(g[tag + '_user_w']).to(DEV).requires_grad_(True)
    uid, pos = T(g[tag + '_uid']).to(DEV), T(g[tag + '_pos']).to(DEV)
    q = torch.nn.functional.embedding(uid, uw)
    loss = (ra.scorer.full_lse(q, iw) - (q * iw[pos]).sum(-1)).mean()
    close(loss.detach().cpu(), g[tag + '_loss'], rtol=1e-5)
    loss.backward()
    close(iw.grad.cpu(), g[tag + '_item_grad'], rtol=1e-4, atol=1e-7)
    close(uw.grad.cpu(), g[tag + '_user_grad'], rtol=1e-4, atol=1e-7)
    # through the retriever
    N, d = iw.shape
    item = torch.nn.Embedding(N, d, padding_idx=0)
    user = torch.nn.Embedding(uw.shape[0], d, padding_idx=0)
    with torch.no_grad():
        item.weight.copy_(T(g[tag + '_item_w']))
        user.weight.copy_(T(g[tag + '_user_w']))
    m = ra.BaseRetriever(None, item_encoder=item, query_encoder=user, scorer=ra.InnerProductScorer(),
                         loss=ra.SoftmaxLoss())
    m.fiid, m.fuid, m.frating = 'item_id', 'user_id', 'rating'
    m.item_fields, m.query_fields = {'item_id'}, {'user_id'}
    m.to(DEV)
    batch = {'user_id': uid, 'item_id': pos, 'rating': torch.ones(len(uid), device=DEV)}
    for fused in (True, False):
        m.zero_grad()
        m.config['train']['fused_full_softmax'] = fused
        l2 = m.training_step(batch)
        close(l2.detach().cpu(), g[tag + '_loss'], rtol=1e-5)
        l2.backward()
        close(m.item_encoder.weight.grad.cpu(), g[tag + '_item_grad'], rtol=1e-4, atol=1e-7)
        close(m.query_encoder.weight.grad.cpu(), g[tag + '_user_grad'], rtol=1e-4, atol=1e-7)


def test_full_softmax_grad_d128(ra):
    """d = 128, N = 5001, B = 300 (ragged vs the 128-query / 32-item tiles): full_lse gradients against
    torch autograd of logsumexp(q @ W[1:].T) on the same device."""
    torch.manual_seed(3)
    N, d, B = 5001, 128, 300
    w = (torch.randn(N, d, device=DEV) * 0.3).requires_grad_(True)
    q = (torch.randn(B, d, device=DEV) * 0.3).requires_grad_(True)
    coef = torch.rand(B, device=DEV)
    (ra.scorer.full_lse(q, w) * coef).sum().backward()
    gw, gq = w.grad.clone(), q.grad.clone()
    w.grad = q.grad = None
    (torch.logsumexp(q.double() @ w[1:].double().t(), -1) * coef.double()).sum().backward()
    assert not gw[0].any()
    close(gq.cpu(), q.grad.cpu(), rtol=1e-4, atol=1e-7)
    close(gw.cpu(), w.grad.cpu(), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('N,d,B', [(5001, 128, 300), (777, 64, 33), (40_000, 32, 129), (2, 128, 5), (100_003, 128, 257),
                                   (3000, 48, 40)])
def test_softmax_recompute_with_query_grad(ra, N, d, B):
    """rsa_fullscore_softmax_dq: the recompute pass that also accumulates d/d query on the matrix cores.  probs are
    bit-equal to the plain recompute pass; query_grad == probs @ items[1:] in float64 (ragged B / N vs the 128-query
    and 32-item tiles, a single item, a padded dim, item-range splits that end inside a tile)."""
    g = torch.Generator(device=DEV).manual_seed(N + d)
    w = torch.empty(N, d, device=DEV).normal_(0, 0.3, generator=g)
    q = torch.empty(B, d, device=DEV).normal_(0, 0.3, generator=g)
    scale = torch.empty(B, device=DEV).uniform_(-1, 1, generator=g)
    lse = ra.ops.fullscore(w, q, want_lse=True)[1]
    plain = ra.ops.fullscore_softmax(w, q, lse, scale)
    probs, gq = ra.ops.fullscore_softmax(w, q, lse, scale, want_query_grad=True)
    assert torch.equal(probs, plain) and gq.shape == (B, d)
    want = plain.double() @ w[1:].double()
    close(gq.cpu(), want.float().cpu(), rtol=1e-4, atol=1e-6)
    # twice the same bits (partials are added in a fixed order)
    assert torch.equal(ra.ops.fullscore_softmax(w, q, lse, scale, want_query_grad=True)[1], gq)


def test_fullscore_config5_shape_properties(ra):
    """BASELINE.json configs[4]: N = 1e6, d = 128, B = 512, k = 100: properties + oracle spot checks."""
    N, d, B, k = 1_000_001, 128, 512, 100
    g = torch.Generator(device=DEV).manual_seed(5)
    iw = torch.empty(N, d, device=DEV).normal_(0, 0.1, generator=g)
    q = torch.empty(B, d, device=DEV).normal_(0, 0.1, generator=g)
    _, lse, tv, ti = ra.ops.fullscore(iw, q, want_lse=True, k=k)
    assert bool((tv[:, :-1] >= tv[:, 1:]).all()) and int(ti.min()) >= 1 and int(ti.max()) < N
    # returned ids carry the returned scores (recomputed by the gather+score kernel), and nothing
    # outside the top-k beats the k-th score on a sample of rows checked exactly on the CPU
    again = ra.ops.fused_forward(iw, q, k, neg_ids=ti)['neg_score']
    close(again.cpu(), tv.cpu(), rtol=1e-4, atol=1e-6)
    rows = [0, 77, 511]
    ref = (q[rows].cpu().double() @ iw[1:].cpu().double().T)
    wv, wi = torch.topk(ref, k)
    close(tv[rows].cpu(), wv.float(), rtol=1e-4, atol=1e-6)
    close(lse[rows].cpu(), torch.logsumexp(ref, -1).float(), rtol=1e-5)
    overlap = [len(set(ti[r].tolist()) & set((wi[i] + 1).tolist())) for i, r in enumerate(rows)]
    assert min(overlap) >= k - 2           # only near-ties at the fp32 noise level may differ
    # linearity: scaling the queries scales the top-k scores and shifts nothing
    _, _, tv2, ti2 = ra.ops.fullscore(iw, q * 2, k=k)
    assert torch.equal(ti2, ti) and torch.equal(tv2, tv * 2)


def test_topk_filter_path_adversarial_catalogs(ra):
    """The threshold-filter top-k must stay exact when the sampled tiles are NOT representative:
    scores increasing with the item id, a few huge outliers hidden between sampled tiles, all-equal
    scores (everything ties), and a zero query."""
    N, d, k = 200_001, 64, 100
    g = torch.Generator().manual_seed(3)
    base = torch.randn(N, d, generator=g) * 0.1
    q = torch.randn(4, d, generator=g)
    cases = {}
    trend = base.clone()
    trend[:, 0] += torch.linspace(-3, 3, N)                  # strong trend along the id axis
    cases['trend'] = (trend, q.clone())
    spikes = base.clone()
    spikes[torch.arange(50, N, 1999)] *= 40.0                 # ~100 outliers
    cases['spikes'] = (spikes, q.clone())
    flat = torch.zeros(N, d)
    flat[:, 0] = 1.0                                          # every item scores the same
    cases['flat'] = (flat, q.clone())
    cases['zero_query'] = (base.clone(), torch.zeros(4, d))
    for name, (iw, qq) in cases.items():
        _, lse, tv, ti = ra.ops.fullscore(iw.to(DEV), qq.to(DEV), want_lse=True, k=k)
        ref = qq.double() @ iw[1:].double().T
        wv, wi = torch.topk(ref, k)
        close(tv.cpu(), wv.float(), rtol=1e-4, atol=1e-5)
        got = ref.gather(1, ti.cpu() - 1)                     # the returned ids carry the returned scores
        close(got.float(), tv.cpu(), rtol=1e-4, atol=1e-5)
        if name in ('flat', 'zero_query'):                    # all ties: the k smallest ids, in order
            assert torch.equal(ti.cpu(), torch.arange(1, k + 1).expand(4, k)), name
        close(lse.cpu(), torch.logsumexp(ref, -1).float(), rtol=1e-5)


def test_topk_large_k_large_batch_trending_scores(ra):
    """B = 2048, k = 500 on a catalog whose scores trend along the id axis (ids sorted by popularity, say):
    most of a query's candidates fall into a few item ranges.  The per-segment lists overflow into the
    query's overflow list instead of sending every row to the exact single-workgroup recompute (which used
    to turn this call from milliseconds into tens of seconds)."""
    import time
    torch.manual_seed(5)
    N, d, B, k = 300_001, 128, 2048, 500
    item = torch.randn(N, d, device=DEV) * 0.1
    item[:, 0] += torch.linspace(3, -3, N, device=DEV)
    q = torch.randn(B, d, device=DEV) * 0.3
    q[:, 0] = q[:, 0].abs() + 0.5                      # every query prefers small ids
    ra.ops.fullscore(item, q, k=k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, lse, tv, ti = ra.ops.fullscore(item, q, want_lse=True, k=k)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ref = q @ item[1:].T
    wv, wi = torch.topk(ref, k)
    close(tv.cpu(), wv.cpu(), rtol=1e-4, atol=1e-5)
    assert (ti == wi + 1).float().mean() > 0.99       # fp32 near-ties may swap neighbours
    close(lse.cpu(), torch.logsumexp(ref.double(), -1).float().cpu(), rtol=1e-5)
    assert elapsed < 0.5, f'top-k fell off the fast path: {elapsed * 1e3:.0f} ms'


def test_fullscore_topk_fuzz(ra):
    """18 random (B, N, d, k) around the dense / filter switch-over (32 768 items) and the tile sizes: logsumexp and
    exact top-k (values; ids wherever fp32 near-ties do not swap neighbours) against a float64 matmul on the device."""
    rs = np.random.RandomState(7)
    for it in range(18):
        d = int(rs.choice([32, 64, 128, 100]))
        N = int(rs.choice([33, 1000, 32768, 32769, 32800, 70001, 200003]))
        B = int(rs.choice([1, 31, 128, 129, 300]))
        k = int(min(N - 1, rs.choice([1, 10, 100, 300])))
        torch.manual_seed(it)
        item = torch.randn(N, d, device=DEV) * 0.3
        q = torch.randn(B, d, device=DEV) * 0.3
        _, lse, tv, ti = ra.ops.fullscore(item, q, want_lse=True, k=k)
        ref = q.double() @ item[1:].double().T
        tag = f'it={it} B={B} N={N} d={d} k={k}'
        np.testing.assert_allclose(lse.cpu(), torch.logsumexp(ref, -1).float().cpu(), rtol=2e-5, err_msg=tag)
        wv, wi = torch.topk(ref, k)
        np.testing.assert_allclose(tv.cpu(), wv.float().cpu(), rtol=1e-4, atol=1e-5, err_msg=tag)
        assert (ti == wi + 1).float().mean() > 0.98, tag
        got = ref.gather(1, ti - 1)                                  # the ids carry the returned values
        np.testing.assert_allclose(got.float().cpu(), tv.cpu(), rtol=1e-4, atol=1e-5, err_msg=tag)
        assert bool((tv[:, :-1] >= tv[:, 1:]).all()), tag            # sorted, descending
