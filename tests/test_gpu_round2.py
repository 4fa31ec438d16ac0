"""GPU parity tests (``-m gpu``) of the round-2 additions: the bucket-line inverse CDF, the SampledSoftmax epilogue of
the fused forward (with the in-forward query gradient), cosine / Euclidean scores over the full catalog, the composed
full-score shapes (embed_dim > 128, k > 1024), the device guard -- each against the CPU oracle (oracle/), the committed
reference fixtures (tests/golden) or torch's own device ops."""
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
    recstudio_amd._native.lib()          # fail loudly if the HIP extension is not there
    assert torch.cuda.is_available()
    torch.cuda.init()
    return recstudio_amd


def rel_close(a, b, rtol=1e-4, atol=1e-6):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


def _tables(N, U, d, seed):
    g = torch.Generator().manual_seed(seed)
    iw = torch.randn(N, d, generator=g) * 0.3
    iw[0] = 0
    uw = torch.randn(U, d, generator=g) * 0.3
    return iw, uw


# --------------------------------------------------------------------------- bucket lines
def test_bucket_lines_golden_and_searchsorted(ra, golden):
    """cdf_lines lookup == the reference fixture ids (edge uniforms included), at the automatic size and at tiny
    forced sizes where most draws take the > 12-entries fallback; and == torch.searchsorted on dense random uniforms
    of a 120 000-item table."""
    g = golden('popular')
    for mode in (0, 1, 2):
        want = np.minimum(g[f'm{mode}_ids'], len(g['counts']) - 1)
        for glog in (None, 2, 4, 7, 12, 20):
            ps = ra.PopularSamplerModel.from_tables(T(g[f'm{mode}_pop_prob']), T(g[f'm{mode}_table']), lookup='lines',
                                                    lines_log2=glog).to(DEV)
            assert ps.cdf_lines.data_ptr() % 128 == 0
            ids, logp = ra.ops.popular_lookup(ps.table, ps.pop_prob, None, 0, T(g[f'm{mode}_u']).to(DEV),
                                              cdf_lines=ps.cdf_lines, lines_log2=ps.lines_log2)
            assert np.array_equal(ids.cpu().numpy(), want)
            rel_close(logp.cpu(), g[f'm{mode}_logp'], rtol=1e-6, atol=1e-7)
    big = ra.PopularSamplerModel(T(g['big_counts']), mode=0, lookup='lines').to(DEV)
    torch.manual_seed(5)
    uu = torch.rand(300_000, device=DEV)
    want = torch.searchsorted(big.table, uu).clamp_(max=big.table.numel() - 1)
    for glog in (None, 8):
        ps = big if glog is None else ra.PopularSamplerModel(T(g['big_counts']), mode=0, lookup='lines', lines_log2=glog).to(DEV)
        ids, logp = ra.ops.popular_lookup(ps.table, ps.pop_prob, None, 0, uu, cdf_lines=ps.cdf_lines, lines_log2=ps.lines_log2)
        assert torch.equal(ids, want)
        rel_close(logp.cpu(), torch.log(ps.pop_prob[want]).cpu(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('n,B', [(64, 100), (256, 9), (100, 33), (1, 300)])
@pytest.mark.parametrize('glog', [None, 6])
def test_fused_forward_sampled_with_bucket_lines(ra, n, B, glog):
    """The fused kernel with cdf_lines draws what torch.rand + torch.searchsorted draw on this device for the same
    seed (one-tile-ahead prefetch for n % 64 == 0, plain lookup otherwise; BPR epilogue with and without the
    in-forward query gradient), and reports log(pop_prob[id])."""
    N, U, d = 50_021, 211, 128
    iw, uw = _tables(N, U, d, n)
    g = torch.Generator().manual_seed(n + B)
    counts = (torch.rand(N, generator=g) ** 5 * 300).long()
    counts[torch.rand(N, generator=g) < 0.4] = 0
    ps = ra.PopularSamplerModel(counts, lookup='lines', lines_log2=glog).to(DEV)
    uid = torch.randint(1, U, (B,), generator=g).to(DEV)
    pos = torch.randint(1, N, (B,), generator=g).to(DEV)
    iwd, uwd = iw.to(DEV), uw.to(DEV)
    torch.manual_seed(11)
    want = torch.searchsorted(ps.table, torch.rand(B, n, device=DEV)).clamp_(max=N - 1)
    variants = [dict()]
    if n % 64 == 0:
        variants += [dict(fused_bpr=True), dict(fused_bpr=True, want_query_grad=True), dict(fused_loss='ssm'),
                     dict(fused_loss='ssm', want_query_grad=True)]
    for kw in variants:
        torch.manual_seed(11)
        o = ra.ops.fused_forward(iwd, uwd, n, query_index=uid, pos_ids=pos, sampler=ra._native.SAMPLER_POPULAR,
                                 **ps.lookup_kwargs(), **kw)
        assert torch.equal(o['neg_ids'], want), kw
        rel_close(o['neg_logp'].cpu(), torch.log(ps.pop_prob[want]).cpu(), rtol=1e-6, atol=1e-7)
        ref = (uwd[uid].unsqueeze(1) * iwd[want]).sum(-1)
        rel_close(o['neg_score'].cpu(), ref.cpu(), rtol=1e-4, atol=1e-5)
    # the Sampler plugin on its own draws the same stream
    torch.manual_seed(11)
    lp, ids, lnp = ps(torch.empty(B, 1, device=DEV), n, pos)
    assert torch.equal(ids, want)
    rel_close(lp.cpu(), torch.log(ps.pop_prob[pos]).cpu(), rtol=1e-6, atol=1e-7)


# --------------------------------------------------------------------------- SampledSoftmax epilogue
@pytest.mark.parametrize('d', [32, 64, 128, 256])
@pytest.mark.parametrize('n', [64, 256, 1024])
@pytest.mark.parametrize('kind', ['given', 'given_logp', 'uniform', 'popular'])
def test_fused_ssm_epilogue_vs_oracle(ra, d, n, kind):
    """fused_loss = 'ssm' (one launch: scores, logsumexp across the n / 64 tiles of a query, loss, d loss/d score,
    d loss/d query) == oracle.sampled_softmax_loss + autograd on the same ids, == the separate loss kernel."""
    if n == 1024 and d in (32, 256):
        pytest.skip('covered by the other dims')
    N, U, B = 3001, 97, 37
    iw, uw = _tables(N, U, d, n + d)
    g = torch.Generator().manual_seed(d + n)
    uid = torch.randint(1, U, (B,), generator=g)
    pos = torch.randint(1, N, (B,), generator=g)
    counts = (torch.rand(N, generator=g) ** 3 * 100).long()
    nat = ra._native
    kw, lpp, lnp = {}, None, None
    if kind.startswith('given'):
        kw = dict(neg_ids=torch.randint(0, N, (B, n), generator=g).to(DEV))
        if kind == 'given_logp':
            lpp = -torch.rand(B, generator=g) * 9
            lnp = -torch.rand(B, n, generator=g) * 9
            kw.update(pos_logp=lpp.to(DEV), neg_logp=lnp.to(DEV))
    elif kind == 'uniform':
        kw = dict(sampler=nat.SAMPLER_UNIFORM)
    else:
        ps = ra.PopularSamplerModel(counts, lookup='lines' if n == 256 else 'auto').to(DEV)
        kw = dict(sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs())
    iwd, uwd = iw.to(DEV), uw.to(DEV)
    torch.manual_seed(21)
    a = ra.ops.fused_forward(iwd, uwd, n, query_index=uid.to(DEV), pos_ids=pos.to(DEV), fused_loss='ssm',
                             want_query_grad=True, **kw)
    torch.manual_seed(21)
    b = ra.ops.fused_forward(iwd, uwd, n, query_index=uid.to(DEV), pos_ids=pos.to(DEV), fused_loss='ssm', **kw)
    torch.manual_seed(21)
    c = ra.ops.fused_forward(iwd, uwd, n, query_index=uid.to(DEV), pos_ids=pos.to(DEV), **kw)      # no epilogue
    ids = a['neg_ids']
    assert torch.equal(ids, b['neg_ids']) and torch.equal(ids, c['neg_ids'])
    if kind == 'popular':
        lpp, lnp = c['pos_logp'].cpu(), c['neg_logp'].cpu()
        rel_close(a['neg_logp'].cpu(), lnp, rtol=1e-6, atol=1e-7)
        rel_close(a['pos_logp'].cpu(), lpp, rtol=1e-6, atol=1e-7)
    # oracle: value + autograd w.r.t. the scores and the query
    q = uw[uid].clone().requires_grad_(True)
    psr = (q * iw[pos]).sum(-1)
    nsr = (q.unsqueeze(1) * iw[ids.cpu()]).sum(-1)
    psr.retain_grad()
    nsr.retain_grad()
    zero_p, zero_n = torch.zeros(B), torch.zeros(B, n)
    val = oracle.sampled_softmax_loss(psr, lpp if lpp is not None else zero_p, nsr, lnp if lnp is not None else zero_n)
    val.backward()
    for o in (a, b):
        rel_close(o['loss'].cpu(), val.detach(), rtol=1e-5)
        rel_close(o['neg_score'].cpu(), nsr.detach(), rtol=1e-4, atol=1e-5)
        rel_close(o['pos_score'].cpu(), psr.detach(), rtol=1e-4, atol=1e-5)
        rel_close(o['dneg'].cpu(), nsr.grad, rtol=1e-4, atol=1e-9)
        rel_close(o['dpos'].cpu(), psr.grad, rtol=1e-4, atol=1e-9)
    rel_close(a['query_grad'].cpu(), q.grad, rtol=2e-4, atol=1e-8)
    # == the stand-alone loss kernel on the unfused scores
    lp_d = None if lpp is None else lpp.to(DEV)
    ln_d = None if lnp is None else lnp.to(DEV)
    loss2, dpos2, dneg2, _ = ra.ops.pairwise_loss(nat.LOSS_SSM, c['pos_score'], c['neg_score'], lp_d, ln_d)
    rel_close(a['loss'].cpu(), loss2.cpu(), rtol=1e-5)
    rel_close(a['dneg'].cpu(), dneg2.cpu(), rtol=1e-4, atol=1e-9)


def test_fused_ssm_extreme_logits_and_padding(ra):
    """Running-maximum rescaling: scores spread over +-60 with log-probabilities down to -18 stay finite and equal the
    oracle; a -inf positive (masked padding) yields NaN loss and gradients for that row like the reference
    (loss_func.py:88-89)."""
    N, d, B, n = 999, 64, 8, 192
    g = torch.Generator().manual_seed(0)
    iw = torch.randn(N, d, generator=g)
    iw[0] = 0
    q = torch.randn(B, d, generator=g) * 3
    pos = torch.randint(1, N, (B,), generator=g)
    neg = torch.randint(1, N, (B, n), generator=g)
    lnp = -torch.rand(B, n, generator=g) * 18
    lpp = -torch.rand(B, generator=g) * 18
    o = ra.ops.fused_forward(iw.to(DEV), q.to(DEV), n, pos_ids=pos.to(DEV), neg_ids=neg.to(DEV), fused_loss='ssm',
                             want_query_grad=True, pos_logp=lpp.to(DEV), neg_logp=lnp.to(DEV))
    qr = q.clone().requires_grad_(True)
    val = oracle.sampled_softmax_loss((qr * iw[pos]).sum(-1), lpp, (qr.unsqueeze(1) * iw[neg]).sum(-1), lnp)
    val.backward()
    assert torch.isfinite(o['loss']).all() and float((qr.unsqueeze(1) * iw[neg]).sum(-1).abs().max()) > 40
    rel_close(o['loss'].cpu(), val.detach(), rtol=1e-5)
    rel_close(o['query_grad'].cpu(), qr.grad, rtol=2e-4, atol=2e-6)      # sums of +-O(1) terms: cancellation
    pos2 = pos.clone()
    pos2[3] = 0
    o = ra.ops.fused_forward(iw.to(DEV), q.to(DEV), n, pos_ids=pos2.to(DEV), neg_ids=neg.to(DEV), fused_loss='ssm',
                             want_query_grad=True, mask_pad_pos=True)
    assert torch.isnan(o['loss']) and torch.isnan(o['row_loss'][3]) and torch.isnan(o['dneg'][3]).all()
    assert torch.isnan(o['query_grad'][3]).all() and torch.isfinite(o['row_loss'][[0, 1, 2, 4]]).all()


@pytest.mark.parametrize('kind', ['uniform', 'popular', 'given'])
def test_fused_ssm_loss_autograd_vs_oracle(ra, kind):
    """fused_ssm_loss through autograd (dense and row-sparse table gradients, write-only backward) == the oracle's
    dense_grads on the drawn ids; the ids are those of the unfused path under the same seed."""
    from recstudio_amd.fused import fused_ssm_loss
    N, U, d, B, n = 2003, 97, 128, 41, 256
    iw, uw = _tables(N, U, d, 5)
    g = torch.Generator().manual_seed(3)
    uid = torch.randint(1, U, (B,), generator=g)
    pos = torch.randint(1, N, (B,), generator=g)
    counts = (torch.rand(N, generator=g) ** 3 * 100).long()
    sampler = {'given': None, 'uniform': ra.UniformSampler(N), 'popular': ra.PopularSamplerModel(counts).to(DEV)}[kind]
    neg_given = torch.randint(0, N, (B, n), generator=g).to(DEV) if kind == 'given' else None
    for sparse in (False, True):
        iwd, uwd = iw.to(DEV).requires_grad_(True), uw.to(DEV).requires_grad_(True)
        torch.manual_seed(77)
        loss, ids = fused_ssm_loss(iwd, uwd, n, query_index=uid.to(DEV), pos_ids=pos.to(DEV), sampler=sampler,
                                   neg_ids=neg_given, sparse_grad=sparse)
        (loss * 2.0).backward()
        lpp = lnp = None
        if kind == 'popular':
            ref = oracle.PopularSamplerModel(counts)
            lpp, lnp = ref.compute_item_p(pos), ref.compute_item_p(ids.cpu())
        val, _, _, gi, gu = oracle.dense_grads(iw, uw, uid, pos, ids.cpu(), loss='ssm',
                                               log_pos_prob=lpp if lpp is not None else torch.zeros(B),
                                               log_neg_prob=lnp if lnp is not None else torch.zeros(B, n))
        rel_close(loss.detach().cpu(), val, rtol=1e-5)
        gi_got = iwd.grad.to_dense() if sparse else iwd.grad
        gu_got = uwd.grad.to_dense() if sparse else uwd.grad
        rel_close(gi_got.cpu(), gi * 2, rtol=3e-4, atol=1e-7)
        rel_close(gu_got.cpu(), gu * 2, rtol=3e-4, atol=1e-7)
        torch.manual_seed(77)
        _, ids2 = ra.retriever_scores(iw.to(DEV), uw.to(DEV), n, query_index=uid.to(DEV), pos_ids=pos.to(DEV),
                                      sampler=sampler, neg_ids=neg_given)
        assert torch.equal(ids, ids2)


# --------------------------------------------------------------------------- cosine / Euclidean over the full catalog
@pytest.mark.parametrize('d', [64, 128])
@pytest.mark.parametrize('name', ['cos', 'euc', 'ip'])
def test_full_catalog_scorers_golden(ra, golden, d, name):
    """([B,D],[N,D]) case of Cosine / Euclidean / InnerProduct scorers (scorer.py:16-34) == the reference fixture."""
    g = golden('score')
    q, items = T(g[f'd{d}_bd_Nd_q']).to(DEV), T(g[f'd{d}_bd_Nd_items']).to(DEV)
    scorer = {'cos': ra.CosineScorer(), 'euc': ra.EuclideanScorer(), 'ip': ra.InnerProductScorer()}[name]
    out = scorer(q, items)
    assert out.shape == (q.shape[0], items.shape[0])
    rel_close(out.cpu(), g[f'd{d}_bd_Nd_{name}'], rtol=1e-4, atol=1e-5)
    if name == 'euc':
        rel_close(ra.NormScorer()(q, items).cpu(), g[f'd{d}_bd_Nd_norm2'], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('mode', ['cos', 'euc'])
@pytest.mark.parametrize('d,N,B,k', [(128, 40_000, 200, 100), (64, 5_000, 33, 10), (100, 70_001, 130, 50)])
def test_full_catalog_cos_euc_scores_lse_topk(ra, mode, d, N, B, k):
    """rsa_fullscore with the cosine / Euclidean tile epilogue: materialised scores, logsumexp and exact top-k (the
    filter path for large catalogs, the dense select for small ones) against the oracle's formulas; gradients of the
    materialised path against torch autograd."""
    g = torch.Generator().manual_seed(d + N)
    iw = torch.randn(N, d, generator=g) * (0.5 + torch.rand(N, 1, generator=g) * 2)
    iw[0] = 0
    q = torch.randn(B, d, generator=g)
    fn = oracle.cosine_score if mode == 'cos' else oracle.euclidean_score
    want = fn(q, iw[1:])
    sm = ra._native.SCORE_COS if mode == 'cos' else ra._native.SCORE_EUC
    scores, lse, _, _ = ra.ops.fullscore(iw.to(DEV), q.to(DEV), want_scores=True, want_lse=True, score_mode=sm)
    rel_close(scores.cpu(), want, rtol=1e-4, atol=1e-4 if mode == 'euc' else 1e-6)
    rel_close(lse.cpu(), torch.logsumexp(want.double(), -1).float(), rtol=1e-5, atol=1e-5)
    _, lse2, tv, ti = ra.ops.fullscore(iw.to(DEV), q.to(DEV), want_lse=True, k=k, score_mode=sm)
    rel_close(lse2.cpu(), lse.cpu(), rtol=1e-6, atol=1e-6)
    wv, wi = torch.topk(want, k)
    rel_close(tv.cpu(), wv, rtol=1e-4, atol=1e-4 if mode == 'euc' else 1e-6)
    # ids: equal wherever the k-th neighbourhood has no near-ties within the fp32 tolerance
    got_scores = torch.gather(want, 1, (ti.cpu() - 1))
    rel_close(got_scores, wv, rtol=1e-4, atol=1e-4 if mode == 'euc' else 1e-6)
    assert (ti.cpu() - 1 == wi).float().mean() > 0.99
    # autograd of the materialised scorer path
    scorer = ra.CosineScorer() if mode == 'cos' else ra.EuclideanScorer()
    qd, xd = q[:16].to(DEV).requires_grad_(True), iw[1:3000].to(DEV).requires_grad_(True)
    w = torch.randn(16, 2999, generator=g).to(DEV)
    (scorer(qd, xd) * w).sum().backward()
    qr, xr = q[:16].clone().requires_grad_(True), iw[1:3000].clone().requires_grad_(True)
    (fn(qr, xr) * w.cpu()).sum().backward()
    rel_close(qd.grad.cpu(), qr.grad, rtol=2e-3, atol=2e-3 if mode == 'euc' else 2e-5)
    rel_close(xd.grad.cpu(), xr.grad, rtol=2e-3, atol=2e-3 if mode == 'euc' else 2e-5)


def test_full_catalog_wide_dim_and_large_k(ra):
    """Shapes outside the single MFMA pass are composed from it: embed_dim 192 / 320 (k split into 128-wide slices)
    and k = 1500 > 1024 (a user whose history + eval.topk exceeds the in-kernel select), incl. topk_mask_history."""
    g = torch.Generator().manual_seed(7)
    for d, N, B, k in ((192, 9_000, 40, 20), (320, 3_000, 17, 5), (64, 6_000, 12, 1500)):
        iw = torch.randn(N, d, generator=g)
        iw[0] = 0
        q = torch.randn(B, d, generator=g)
        want = q @ iw[1:].t()
        scores, lse, tv, ti = ra.ops.fullscore(iw.to(DEV), q.to(DEV), want_scores=True, want_lse=True, k=k)
        rel_close(scores.cpu(), want, rtol=1e-4, atol=1e-4)
        rel_close(lse.cpu(), torch.logsumexp(want, -1), rtol=1e-5, atol=1e-4)
        wv, wi = torch.topk(want, k)
        rel_close(tv.cpu(), wv, rtol=1e-4, atol=1e-4)
        assert (ti.cpu() - 1 == wi).float().mean() > 0.99
        for mode, fn in ((ra._native.SCORE_COS, oracle.cosine_score), (ra._native.SCORE_EUC, oracle.euclidean_score)):
            s2 = ra.ops.fullscore(iw.to(DEV), q.to(DEV), want_scores=True, score_mode=mode)[0]
            rel_close(s2.cpu(), fn(q, iw[1:]), rtol=1e-4, atol=1e-3 if mode == 2 else 1e-6)


def test_cosine_retriever_trains_and_evaluates(ra, golden):
    """BaseRetriever(scorer=CosineScorer()) used to train and then fail at its first validation step: topk() now runs
    on the MFMA kernel's cosine epilogue; ids == the oracle's topk_with_history(cosine=True) on the fixture."""
    g = golden('topk')
    iw, uw = T(g['item_w']), T(g['user_w'])
    uid, hist = T(g['uid']), T(g['hist'])
    for scorer, cos in ((ra.CosineScorer(), True), (ra.InnerProductScorer(), False)):
        m = ra.BaseRetriever({'train': {'seed': 1}}, scorer=scorer,
                             item_encoder=torch.nn.Embedding(iw.shape[0], iw.shape[1], padding_idx=0),
                             query_encoder=torch.nn.Embedding(uw.shape[0], uw.shape[1], padding_idx=0))
        m.fiid, m.fuid, m.frating = 'item_id', 'user_id', 'rating'
        m.fields = {'item_id', 'user_id', 'rating'}
        m.item_fields, m.query_fields = {'item_id'}, {'user_id'}
        with torch.no_grad():
            m.item_encoder.weight.copy_(iw)
            m.query_encoder.weight.copy_(uw)
        m.to(DEV)
        m._update_item_vector()
        score, items = m.topk({'user_id': uid.to(DEV)}, 10, hist.to(DEV))
        ws, wi = oracle.topk_with_history(uw[uid], iw, 10, hist, cosine=cos)
        rel_close(score.cpu(), ws, rtol=1e-4, atol=1e-6)
        assert torch.equal(items.cpu(), wi)
        if not cos:
            assert torch.equal(items.cpu(), T(g['items']))


# --------------------------------------------------------------------------- hygiene
def test_reduce_scratch_is_per_stream_and_library_allocates_nothing(ra):
    """The in-kernel loss reduction uses the caller's scratch: two streams get two blocks, repeated launches leave the
    arrival counter at zero, and the loss equals the mean of the row losses."""
    N, U, d, B, n = 2003, 97, 128, 300, 64
    iw, uw = _tables(N, U, d, 1)
    iwd, uwd = iw.to(DEV), uw.to(DEV)
    uid = torch.randint(1, U, (B,), device=DEV)
    pos = torch.randint(1, N, (B,), device=DEV)
    neg = torch.randint(1, N, (B, n), device=DEV)
    outs = []
    streams = [torch.cuda.current_stream(), torch.cuda.Stream()]
    for s in streams:
        with torch.cuda.stream(s):
            for _ in range(3):
                o = ra.ops.fused_forward(iwd, uwd, n, query_index=uid, pos_ids=pos, neg_ids=neg, fused_bpr=True)
            outs.append(o)
    torch.cuda.synchronize()
    keys = {k for k in ra.ops._SCRATCH if k[0] == torch.cuda.current_device()}
    assert len({k[1] for k in keys}) >= 2
    for o in outs:
        rel_close(o['loss'].cpu(), o['row_loss'].mean().cpu(), rtol=1e-5)
    for k in keys:
        assert int(ra.ops._SCRATCH[k][:4].view(torch.int32)[0]) == 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs')
def test_ops_follow_the_tensors_device(ra):
    """Tensors on cuda:1 while cuda:0 is current: the launch runs on device 1's stream with device 1's scratch and
    generator; mixing devices raises."""
    torch.cuda.set_device(0)
    N, U, d, B, n = 2003, 97, 128, 50, 64
    iw, uw = _tables(N, U, d, 1)
    d1 = torch.device('cuda', 1)
    uid = torch.randint(1, U, (B,))
    pos = torch.randint(1, N, (B,))
    torch.manual_seed(5)
    o1 = ra.ops.fused_forward(iw.to(d1), uw.to(d1), n, query_index=uid.to(d1), pos_ids=pos.to(d1),
                              sampler=ra._native.SAMPLER_UNIFORM, fused_bpr=True)
    torch.manual_seed(5)
    o0 = ra.ops.fused_forward(iw.to(DEV), uw.to(DEV), n, query_index=uid.to(DEV), pos_ids=pos.to(DEV),
                              sampler=ra._native.SAMPLER_UNIFORM, fused_bpr=True)
    assert o1['neg_ids'].device == d1 and torch.equal(o1['neg_ids'].cpu(), o0['neg_ids'].cpu())
    rel_close(o1['loss'].cpu(), o0['loss'].cpu(), rtol=1e-6)
    with pytest.raises(RuntimeError, match='different devices'):
        ra.ops.embedding_gather(iw.to(d1), uid.to(DEV))


# --------------------------------------------------------------------------- BASELINE.json configs[2] / configs[3] as configs
def test_config2_sasrec_d128_L50_ssm_n256_training_step(ra, golden):
    """configs[2] composed: SASRec (SeqDataset, max_seq_len = 50) at d = 128 with SampledSoftmaxLoss + PopularSamplerModel,
    n = 256.  One training_step on a fixed batch (single fused launch for sampling + scores + loss + d loss/d query)
    against the reference's op sequence restated on the CPU (sasrec.py:37-67 query encoder with stock torch modules,
    baseretriever.py:153-171 scores, loss_func.py:80-90): loss and the gradients of the item table (which also feeds the
    history gather) and of the position embedding."""
    import copy
    from test_dataset_golden import make
    g = golden('data_ml100k')
    sq = make(ra.SeqDataset, g, max_seq_len=50)
    trn, _, _ = sq.build(split_ratio=2)
    n = 256
    cfg = {'model': {'embed_dim': 128, 'dropout_rate': 0.0}, 'train': {'negative_count': n, 'batch_size': 512,
                                                                       'init_method': 'normal', 'seed': 7}}
    model = ra.SASRec(cfg, loss=ra.SampledSoftmaxLoss(), sampler=ra.PopularSamplerModel(trn.item_freq))
    model._init_model(trn)
    model._init_parameter()
    with torch.no_grad():                       # the 0.02 init gives near-constant scores: spread them
        model.item_encoder.weight.mul_(15.0)
        model.item_encoder.weight[0] = 0
    model.to(DEV)
    model.train()
    batch = next(iter(trn.train_loader(batch_size=512, shuffle=False)))
    assert batch['in_item_id'].shape[1] <= 50 and model.item_encoder.weight.shape[1] == 128
    ref = copy.deepcopy(model).cpu()
    bd = {k: v.to(DEV) for k, v in batch.items()}
    torch.manual_seed(123)
    loss = model.training_step(bd)
    loss.backward()
    # the same negatives through the sampler plugin under the same seed (same device stream)
    torch.manual_seed(123)
    lpp, neg, lnp = model.sampler(torch.empty(len(batch['item_id']), 1, device=DEV), n, bd['item_id'])
    # reference op sequence on the CPU
    enc = ref.query_encoder
    hist = batch['in_item_id']
    B, L = hist.shape
    positions = torch.arange(L).unsqueeze(0).expand(B, L)
    seq = torch.nn.functional.embedding(hist, ref.item_encoder.weight, padding_idx=0) + enc.position_emb(positions)
    causal = torch.triu(torch.ones(L, L, dtype=torch.bool), 1)
    out = enc.transformer_layer(enc.dropout(seq), mask=causal, src_key_padding_mask=hist == 0)
    last = (batch['seqlen'] - 1).clamp(min=0).view(-1, 1, 1).expand(-1, 1, out.shape[-1])
    q = out.gather(1, last).squeeze(1)
    w = ref.item_encoder.weight
    pos_s = (q * torch.nn.functional.embedding(batch['item_id'], w, padding_idx=0)).sum(-1)
    neg_s = (q.unsqueeze(1) * torch.nn.functional.embedding(neg.cpu(), w, padding_idx=0)).sum(-1)
    want = oracle.sampled_softmax_loss(pos_s, lpp.cpu(), neg_s, lnp.cpu())
    want.backward()
    assert float(neg_s.std()) > 0.05
    rel_close(loss.detach().cpu(), want.detach(), rtol=2e-4)
    gi, gi_ref = model.item_encoder.weight.grad.cpu(), ref.item_encoder.weight.grad
    assert not gi[0].any() and torch.isfinite(gi).all()
    scale = float(gi_ref.abs().max())
    rel_close(gi, gi_ref, rtol=2e-3, atol=2e-4 * scale)
    gp, gp_ref = model.query_encoder.position_emb.weight.grad.cpu(), enc.position_emb.weight.grad
    rel_close(gp, gp_ref, rtol=2e-3, atol=2e-4 * float(gp_ref.abs().max()))
    # The device loader hands the tower the CSR view of the interaction column (VERDICT r2 missing #3): the
    # same step through rsa_seg_gather's rows form gives the same loss and gradients -- and the item-table gradient
    # (history gather + scores) is bit-equal run to run: the backward of the gather is the sorted, atomics-free scatter
    dl = trn.device_train_loader(512, shuffle=False, device=DEV)
    db = next(iter(dl))
    assert '_seg' in db and torch.equal(db['in_item_id'], bd['in_item_id']) and torch.equal(db['item_id'], bd['item_id'])
    flat, st, en = db['_seg']
    assert torch.equal(ra.ops.seg_gather(model.item_encoder.weight.detach(), flat, st, en, db['in_item_id'].shape[1], want_ids=False)[1],
                       ra.ops.embedding_gather(model.item_encoder.weight.detach(), db['in_item_id']))
    grads = []
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        torch.manual_seed(123)
        l2 = model.training_step(dict(db))
        l2.backward()
        grads.append(model.item_encoder.weight.grad.clone())
    rel_close(l2.detach().cpu(), loss.detach().cpu(), rtol=1e-6)
    rel_close(grads[0].cpu(), grads[1].cpu(), rtol=1e-5, atol=1e-6 * scale)         # (the stock Transformer's backward sits in between)
    rel_close(grads[0].cpu(), gi, rtol=1e-4, atol=1e-5 * scale)
    from recstudio_amd.retriever import _embedding_grad
    gseq = torch.randn(*db['in_item_id'].shape, 128, device=DEV)
    ga, gb = (_embedding_grad(gseq, db['in_item_id'], model.item_encoder.weight.shape[0]) for _ in range(2))
    assert torch.equal(ga, gb) and not ga[0].any()                                  # the gather's own backward: bit-reproducible
    want_g = torch.zeros_like(ga).index_add_(0, db['in_item_id'].reshape(-1), gseq.reshape(-1, 128))
    want_g[0] = 0
    rel_close(ga.cpu(), want_g.cpu(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('layout', ['block', 'interleaved'])
def test_config3_per_gpu_shape_properties(ra, layout):
    """configs[3] at the per-GPU shape of an 8-way shard: 12.5 M-row block (6.4 GB), n = 1024, B = 4096.
    The negatives drawn INSIDE the routing launch are bit-exact vs torch.randint over the GLOBAL id range (this rank's
    rows of the job-wide call), every element lands once in its owner's fixed-capacity segment without overflow at the
    default slack, the segment form of the scoring kernel on the block is spot-checked against gathered rows, and the
    scores find their way home through the slots (BPR loss of the home kernel == the stand-alone loss kernel)."""
    from recstudio_amd.shard import HipBackend, RowShardPlan
    n_global, world, B, n, d, rank = 100_000_001, 8, 4096, 1024, 128, 3
    plan = RowShardPlan(n_global, world, layout=layout)          # contiguous 12.5 M-row blocks, or rows r, r + 8, ...
    rows = plan.rows_per_shard
    assert rows == 12_500_001 and plan.n_local(0) == rows
    gen = torch.Generator(device=DEV).manual_seed(1)
    block = torch.empty(rows, d, device=DEV).normal_(0, 0.02, generator=gen)
    q_all = torch.empty(world * B, d, device=DEV).normal_(0, 0.02, generator=gen)
    pos = torch.randint(1, n_global, (B,), device=DEV, generator=gen)
    hb = HipBackend()
    st = hb.new_state(DEV)
    spec = hb.sampler_spec(ra.UniformSampler(n_global))
    g1 = torch.Generator(device=DEV).manual_seed(31)
    counts = hb.sample_route(st, plan, rank, pos, n, 1, 0, spec, g1, count_only=True).cpu().long()
    assert g1.get_offset() == 0 and int(counts.sum()) == B * (n + 1)
    cap = (int(counts.max() * 1.08) + 4096 + 255) // 256 * 256
    r = hb.sample_route(st, plan, rank, pos, n, 1, cap, spec, g1, want_ids=True)
    # the ids: rows [rank*B, (rank+1)*B) of ONE torch.randint call over [world*B, n] from the same generator state
    g2 = torch.Generator(device=DEV).manual_seed(31)
    want = torch.randint(1, n_global, (world * B, n), device=DEV, generator=g2)[rank * B:(rank + 1) * B]
    assert torch.equal(r['neg_ids'], want) and g1.get_offset() == g2.get_offset()
    ids_flat = torch.cat([pos.view(-1, 1), r['neg_ids']], 1).reshape(-1)
    assert torch.equal(counts, torch.bincount(plan.owner(ids_flat).cpu(), minlength=world))
    stride = r['stride']
    assert stride == cap + hb.HDR
    send, slot_of = r['send'].view(world, stride), r['slot_of'].long()
    assert torch.equal(send[:, 0].cpu(), counts) and int(send[:, 1].sum()) == 0        # headers: live counts, nothing dropped
    assert (slot_of >= 0).all() and slot_of.unique().numel() == B * (n + 1)
    keys = r['send'][slot_of]
    assert torch.equal(keys & 0xffffffff, plan.local(ids_flat)) and torch.equal(slot_of // stride, plan.owner(ids_flat))
    assert torch.equal(keys >> 32, (rank * B + torch.arange(B, device=DEV)).repeat_interleave(n + 1))
    # owner side: this rank plays owner 5 (rows [5 * rows, 6 * rows)); as if all 8 sources had sent this same segment
    seg = send[5].contiguous()
    recv = seg.repeat(world)
    flags = hb.new_state(DEV)
    sc = hb.score_segments(flags, block, q_all, recv, world, stride).view(world, stride)
    assert int(flags['step_dropped']) == 0 and int(flags['overflow']) == 0
    live = int(counts[5])
    pick = hb.HDR + torch.randint(0, live, (2000,), device=DEV)
    k = seg[pick]
    ref = (block[k & 0xffffffff] * q_all[k >> 32]).sum(-1)
    rel_close(sc[0][pick].cpu(), ref.cpu(), rtol=1e-4, atol=1e-7)
    assert torch.equal(sc[0][hb.HDR:hb.HDR + live], sc[7][hb.HDR:hb.HDR + live])
    # home side: stand-in scores (local row number, exact in fp32 below 2^24) come home through the slots; the fused
    # BPR loss + gradients == the stand-alone loss kernel on the gathered scores
    fake = (r['send'] & 0xffffffff).float()
    out = hb.home(fake, r['slot_of'], B, n)
    want_home = plan.local(ids_flat).view(B, n + 1).float()
    assert torch.equal(out['pos_score'], want_home[:, 0]) and torch.equal(out['neg_score'], want_home[:, 1:])
    sc_rand = torch.randn(world * stride, device=DEV)
    o1 = hb.home(sc_rand, r['slot_of'], B, n)
    o2 = hb.home(sc_rand, r['slot_of'], B, n, loss='bpr', want_grad=True, want_dsend=True)
    loss, dpos, dneg, _ = ra.ops.pairwise_loss(ra._native.LOSS_BPR, o1['pos_score'], o1['neg_score'])
    rel_close(o2['loss'].cpu(), loss.cpu(), rtol=1e-5)
    rel_close(o2['dneg'].cpu(), dneg.cpu(), rtol=1e-4, atol=1e-12)
    rel_close(o2['dpos'].cpu(), dpos.cpu(), rtol=1e-4, atol=1e-12)
    d_flat = torch.cat([o2['dpos'].view(-1, 1), o2['dneg']], 1).reshape(-1)
    assert torch.equal(o2['d_send'][slot_of], d_flat)
    assert torch.equal(hb.scatter_slots(o2['dpos'], o2['dneg'], r['slot_of'], world * stride)[slot_of], d_flat)


def test_softmax_loss_second_branch_golden(ra, golden):
    """SoftmaxLoss with pos_score [B, L] and all_score [B, N] of the same rank (loss_func.py:43-47): value and both
    gradients == the reference fixture (padded -inf positives dropped from the row mean)."""
    g = golden('loss')
    pos = T(g['softmax_multi_pad_pos_score']).to(DEV).requires_grad_(True)
    alls = T(g['softmax_multi_pad_all_score']).to(DEV).requires_grad_(True)
    val = ra.SoftmaxLoss()(None, pos, alls)
    val.backward()
    rel_close(val.detach().cpu(), g['softmax_multi_pad_loss'], rtol=1e-5)
    rel_close(alls.grad.cpu(), g['softmax_multi_pad_grad_all_score'], rtol=1e-4, atol=1e-8)
    rel_close(pos.grad.cpu(), g['softmax_multi_pad_grad_pos_score'], rtol=1e-4, atol=1e-8)
    # first branch unchanged
    pos1 = T(g['softmax_full_pos_score']).to(DEV).requires_grad_(True)
    all1 = T(g['softmax_full_all_score']).to(DEV).requires_grad_(True)
    v1 = ra.SoftmaxLoss()(None, pos1, all1)
    v1.backward()
    rel_close(v1.detach().cpu(), g['softmax_full_loss'], rtol=1e-5)
    rel_close(all1.grad.cpu(), g['softmax_full_grad_all_score'], rtol=1e-4, atol=1e-8)


def test_config2_bench_shape_properties(ra):
    """configs[2] at the bench's size (N = 1e6 + 1, d = 128, B = 8192 prefixes, n = 256, popularity sampler, fused
    SampledSoftmax with the in-forward query gradient): size-independent properties + an oracle spot check."""
    N, d, B, n = 1_000_001, 128, 8192, 256
    g = torch.Generator(device=DEV).manual_seed(1)
    iw = torch.empty(N, d, device=DEV).normal_(0, 0.3, generator=g)
    iw[0] = 0
    q = torch.empty(B, d, device=DEV).normal_(0, 0.3, generator=g)
    counts = (torch.rand(N, generator=torch.Generator().manual_seed(2)) ** 8 * 1e4).long()
    seen = torch.nonzero(counts[1:] > 0).flatten() + 1          # positives are items that occur (log-prob finite)
    pos = seen[torch.randint(0, seen.numel(), (B,), generator=torch.Generator().manual_seed(3))].to(DEV)
    ps = ra.PopularSamplerModel(counts).to(DEV)
    assert ps.cdf_lines is not None
    kw = dict(pos_ids=pos, sampler=ra._native.SAMPLER_POPULAR, **ps.lookup_kwargs())
    torch.manual_seed(2022)
    o = ra.ops.fused_forward(iw, q, n, fused_loss='ssm', want_query_grad=True, **kw)
    # (1) ids == the reference's op sequence on this device; log-probs of those ids
    torch.manual_seed(2022)
    want = torch.searchsorted(ps.table, torch.rand(B, n, device=DEV)).clamp_(max=N - 1)
    assert torch.equal(o['neg_ids'], want)
    rel_close(o['neg_logp'].cpu(), torch.log(ps.pop_prob[want]).cpu(), rtol=1e-6, atol=1e-7)
    # (2) the mean reduced in the kernel == mean of the row losses; softmax sums to one: sum_j dneg + dpos == 0
    rel_close(o['loss'].cpu(), o['row_loss'].double().mean().float().cpu(), rtol=1e-6)
    assert float((o['dneg'].double().sum(1) + o['dpos'].double()).abs().max()) < 1e-9
    assert float(o['dneg'].min()) >= 0 and float(o['dpos'].max()) <= 0
    # (3) idempotence: the same ids given (with their log-probs) reproduce loss and gradients bit for bit
    again = ra.ops.fused_forward(iw, q, n, pos_ids=pos, neg_ids=o['neg_ids'], fused_loss='ssm', want_query_grad=True,
                                 pos_logp=o['pos_logp'], neg_logp=o['neg_logp'])
    assert torch.equal(again['loss'], o['loss']) and torch.equal(again['dneg'], o['dneg'])
    assert torch.equal(again['query_grad'], o['query_grad'])
    # (4) == the two-launch path (scores, then rsa_pairwise_loss)
    torch.manual_seed(2022)
    c = ra.ops.fused_forward(iw, q, n, **kw)
    loss2, dpos2, dneg2, _ = ra.ops.pairwise_loss(ra._native.LOSS_SSM, c['pos_score'], c['neg_score'], c['pos_logp'], c['neg_logp'])
    rel_close(o['loss'].cpu(), loss2.cpu(), rtol=1e-5)
    rel_close(o['dneg'].cpu(), dneg2.cpu(), rtol=1e-4, atol=1e-12)
    # (5) oracle spot check on 48 whole rows: loss terms and d loss/d query
    sel = torch.randint(0, B, (48,))
    qs = q[sel.to(DEV)].cpu().clone().requires_grad_(True)
    ids = o['neg_ids'][sel.to(DEV)].cpu()
    psr = (qs * iw[pos[sel.to(DEV)]].cpu()).sum(-1)
    nsr = (qs.unsqueeze(1) * iw[ids.to(DEV)].cpu()).sum(-1)
    z = torch.cat([(psr - o['pos_logp'][sel.to(DEV)].cpu()).unsqueeze(1), nsr - o['neg_logp'][sel.to(DEV)].cpu()], 1)
    rows = torch.logsumexp(z, 1) - z[:, 0]
    rel_close(o['row_loss'][sel.to(DEV)].cpu(), rows.detach(), rtol=1e-4, atol=1e-6)
    (rows.sum() / B).backward()
    rel_close(o['query_grad'][sel.to(DEV)].cpu(), qs.grad, rtol=3e-4, atol=1e-9)


def test_config1_headline_shape_fused_bpr_properties(ra):
    """configs[1] at the bench's batch (B = 65 536, n = 64, N = 1e7 + 1, popularity sampler through the bucket lines):
    the single-launch step's ids == torch's, loss == the separate loss kernel's on the same scores, the in-kernel mean ==
    mean(row_loss), and the in-forward query gradient == the backward kernel's."""
    N, U, d, B, n = 10_000_001, 1_000_001, 128, 65536, 64
    g = torch.Generator(device=DEV).manual_seed(1)
    iw = torch.empty(N, d, device=DEV).normal_(0, 0.1, generator=g)
    iw[0] = 0
    uw = torch.empty(U, d, device=DEV).normal_(0, 0.1, generator=g)
    counts = (torch.rand(N, generator=torch.Generator().manual_seed(2)) ** 8 * 1e4).long()
    ps = ra.PopularSamplerModel(counts).to(DEV)
    uid = torch.randint(1, U, (B,), device=DEV, generator=g)
    pos = torch.randint(1, N, (B,), device=DEV, generator=g)
    kw = dict(query_index=uid, pos_ids=pos, sampler=ra._native.SAMPLER_POPULAR, **ps.lookup_kwargs())
    torch.manual_seed(5)
    o = ra.ops.fused_forward(iw, uw, n, fused_bpr=True, want_query_grad=True, **kw)
    torch.manual_seed(5)
    want = torch.searchsorted(ps.table, torch.rand(B, n, device=DEV)).clamp_(max=N - 1)
    assert torch.equal(o['neg_ids'], want)
    rel_close(o['loss'].cpu(), o['row_loss'].double().mean().float().cpu(), rtol=1e-6)
    loss2, dpos2, dneg2, _ = ra.ops.pairwise_loss(ra._native.LOSS_BPR, o['pos_score'], o['neg_score'])
    rel_close(o['loss'].cpu(), loss2.cpu(), rtol=1e-5)
    rel_close(o['dneg'].cpu(), dneg2.cpu(), rtol=1e-4, atol=1e-12)
    _, _, qg = ra.ops.fused_backward(iw, uw, o['neg_ids'], o['dneg'], query_index=uid, pos_ids=pos, dpos=o['dpos'],
                                     dense_item_grad=False, row_item_grad=False, want_query_grad=True)
    rel_close(o['query_grad'].cpu(), qg.cpu(), rtol=3e-4, atol=1e-10)
    # run-to-run reproducibility of the in-kernel reduction (fixed-point, order-independent)
    torch.manual_seed(5)
    o2 = ra.ops.fused_forward(iw, uw, n, fused_bpr=True, want_query_grad=True, **kw)
    assert torch.equal(o2['loss'], o['loss'])


def test_in_kernel_mean_is_batch_size_independent(ra):
    """The fixed-point mean reduction holds each workgroup's share of the MEAN: large batches with large row losses
    (SampledSoftmax over 1024 negatives: ~ln 1025 per row, 200 000 rows) neither overflow nor lose precision, and a
    single -inf positive turns the mean into NaN like the reference."""
    N, d, B, n = 20_001, 64, 200_000, 1024
    g = torch.Generator(device=DEV).manual_seed(0)
    iw = torch.empty(N, d, device=DEV).normal_(0, 0.05, generator=g)
    q = torch.empty(B, d, device=DEV).normal_(0, 0.05, generator=g)
    pos = torch.randint(1, N, (B,), device=DEV, generator=g)
    torch.manual_seed(3)
    o = ra.ops.fused_forward(iw, q, n, pos_ids=pos, sampler=ra._native.SAMPLER_UNIFORM, fused_loss='ssm')
    want = o['row_loss'].double().mean()
    assert 6.5 < float(want) < 7.5
    rel_close(o['loss'].cpu(), want.float().cpu(), rtol=2e-7)
    pos[12345] = 0
    o = ra.ops.fused_forward(iw, q, n, pos_ids=pos, sampler=ra._native.SAMPLER_UNIFORM, fused_loss='ssm', mask_pad_pos=True)
    assert torch.isnan(o['loss']) and int(torch.isnan(o['row_loss']).sum()) == 1
    pos[12345] = 7
    o = ra.ops.fused_forward(iw, q, n, pos_ids=pos, sampler=ra._native.SAMPLER_UNIFORM, fused_loss='ssm', mask_pad_pos=True)
    assert torch.isfinite(o['loss'])                     # the flag word was cleared by the launch that read it


def test_retriever_topk_with_history_longer_than_1024(ra):
    """BaseRetriever.topk asks the scorer for k + |history| candidates (baseretriever.py:384): a user whose train + val
    history is longer than the in-kernel select's 1024 (ml-1m: ~1.8 k) used to fail; the composed path == the oracle."""
    N, U, d, B, k, Lh = 6001, 50, 64, 9, 100, 1300
    g = torch.Generator().manual_seed(0)
    iw = torch.randn(N, d, generator=g)
    iw[0] = 0
    uw = torch.randn(U, d, generator=g)
    hist = torch.zeros(B, Lh, dtype=torch.int64)
    for b in range(B):
        m = int(torch.randint(5, Lh + 1, (1,), generator=g))
        hist[b, :m] = torch.randperm(N - 1, generator=g)[:m] + 1
    uid = torch.randint(1, U, (B,), generator=g)
    # adversarial rows: the history IS the head of the ranking (rows 0 and 1: their 1000 / 1290 best items), so fewer than
    # k of the 1024 in-kernel candidates survive and the row takes the wide path
    for b, m_top in ((0, 1000), (1, 1290)):
        best = torch.topk(uw[uid[b]] @ iw[1:].t(), m_top).indices + 1
        hist[b] = 0
        hist[b, :m_top] = best
    m = ra.BaseRetriever({'train': {'seed': 1}}, scorer=ra.InnerProductScorer(),
                         item_encoder=torch.nn.Embedding(N, d, padding_idx=0),
                         query_encoder=torch.nn.Embedding(U, d, padding_idx=0))
    m.fiid, m.fuid, m.frating = 'item_id', 'user_id', 'rating'
    m.item_fields, m.query_fields = {'item_id'}, {'user_id'}
    with torch.no_grad():
        m.item_encoder.weight.copy_(iw)
        m.query_encoder.weight.copy_(uw)
    m.to(DEV)
    m._update_item_vector()
    score, items = m.topk({'user_id': uid.to(DEV)}, k, hist.to(DEV))
    ws, wi = oracle.topk_with_history(uw[uid], iw, k, hist)
    rel_close(score.cpu(), ws, rtol=1e-4, atol=1e-5)
    assert (items.cpu() == wi).float().mean() > 0.995          # fp32 near-ties may swap neighbours
    for b in range(B):
        assert not set(items[b].tolist()) & set(hist[b][hist[b] > 0].tolist())


# --------------------------------------------------------------------------- BPR epilogue, queries longer than one tile
@pytest.mark.parametrize('d', [32, 64, 128, 256, 48])
@pytest.mark.parametrize('n,B', [(128, 301), (192, 64), (1024, 70), (448, 3)])
def test_fused_bpr_long_queries_deterministic_and_exact(ra, d, n, B):
    """fused_loss = BPR with num_neg = 64 * T, T > 1: a query's tiles are walked by the waves of ONE
    workgroup and their partial sums meet in LDS in wave order -- no float atomics, no memsets -- so row_loss, dpos, the
    in-forward query gradient and the loss are bit-equal run to run, and equal the separate loss kernel / the oracle.
    d = 48 is outside the walk kernel's dims: the wrapper runs the loss as its own launch (same contract)."""
    N, U = 20_011, 97
    iw, uw = _tables(N, U, d, n + d)
    iwd, uwd = iw.to(DEV), uw.to(DEV)
    g = torch.Generator().manual_seed(B)
    uid = torch.randint(1, U, (B,), generator=g).to(DEV)
    pos = torch.randint(0, N, (B,), generator=g).to(DEV)           # a few padded positives (id 0)
    neg = torch.randint(1, N, (B, n), generator=g).to(DEV)
    qg = d != 48
    runs = [ra.ops.fused_forward(iwd, uwd, n, query_index=uid, pos_ids=pos, neg_ids=neg, fused_bpr=True,
                                 want_query_grad=qg) for _ in range(3)]
    keys = ['loss', 'row_loss', 'dpos', 'dneg', 'neg_score', 'pos_score'] + (['query_grad'] if qg else [])
    for o in runs[1:]:
        for k in keys:
            assert torch.equal(o[k], runs[0][k]), k
    o = runs[0]
    ps, ns = oracle.retriever_forward(iw, uw[uid.cpu()], pos.cpu(), neg.cpu())
    rel_close(o['pos_score'].cpu(), ps, rtol=1e-4, atol=1e-5)
    rel_close(o['neg_score'].cpu(), ns, rtol=1e-4, atol=1e-5)
    psr, nsr = ps.clone().requires_grad_(True), ns.clone().requires_grad_(True)
    want = oracle.bpr_loss(psr, nsr)
    want.backward()
    rel_close(o['loss'].cpu(), want.detach(), rtol=1e-5)
    rel_close(o['dneg'].cpu(), nsr.grad, rtol=1e-4, atol=1e-9)
    rel_close(o['dpos'].cpu(), psr.grad, rtol=1e-4, atol=1e-9)
    rel_close(o['row_loss'].mean().cpu(), want.detach(), rtol=1e-5)
    if qg:
        want_q = psr.grad.unsqueeze(1) * iw[pos.cpu()] + (nsr.grad.unsqueeze(-1) * iw[neg.cpu()]).sum(1)
        rel_close(o['query_grad'].cpu(), want_q, rtol=2e-4, atol=1e-8)
    # in-kernel samplers through the same walk: same ids as the one-tile kernel's stream, forward-only == training forward
    if d == 128:
        torch.manual_seed(3)
        a = ra.ops.fused_forward(iwd, uwd, n, query_index=uid, pos_ids=pos, sampler=ra._native.SAMPLER_UNIFORM, fused_bpr=True)
        torch.manual_seed(3)
        b = ra.ops.fused_forward(iwd, uwd, n, query_index=uid, pos_ids=pos, sampler=ra._native.SAMPLER_UNIFORM, fused_bpr=True,
                                 want_query_grad=True)
        torch.manual_seed(3)
        want_ids = torch.randint(1, N, (B, n), device=DEV)
        assert torch.equal(a['neg_ids'], want_ids) and torch.equal(b['neg_ids'], want_ids)
        rel_close(a['loss'].cpu(), b['loss'].cpu(), rtol=1e-6)
        rel_close(a['neg_score'].cpu(), b['neg_score'].cpu(), rtol=1e-5, atol=1e-6)


def test_fused_step_object_replays_the_call_with_fresh_draws(ra):
    """ops.FusedStep: the frozen call draws what successive fused_forward calls draw (generator advanced alike) and
    writes the same outputs into its own buffers."""
    N, U, d, B, n = 5003, 97, 128, 70, 64
    iw, uw = _tables(N, U, d, 3)
    iwd, uwd = iw.to(DEV), uw.to(DEV)
    uid = torch.randint(1, U, (B,), device=DEV)
    pos = torch.randint(1, N, (B,), device=DEV)
    kw = dict(query_index=uid, pos_ids=pos, sampler=ra._native.SAMPLER_UNIFORM, fused_bpr=True)
    torch.manual_seed(9)
    want = [ra.ops.fused_forward(iwd, uwd, n, **kw) for _ in range(3)]
    torch.manual_seed(9)
    step = ra.ops.FusedStep(iwd, uwd, n, **kw)
    for k in range(3):
        o = step() if k else step.out
        assert torch.equal(o['neg_ids'], want[k]['neg_ids']) and torch.equal(o['loss'], want[k]['loss'])
        assert torch.equal(o['dneg'], want[k]['dneg'])
    assert torch.cuda.default_generators[0].get_offset() == 3 * ra.rng.counter_offset(B * n, ra.rng.grid_threads(B * n, *ra.rng.device_props(DEV)), 4)


# --------------------------------------------------------------------------- in-forward SGD for rows one element owns
@pytest.mark.parametrize('N,d,B,kind', [(200_003, 128, 700, 'uniform'), (3001, 64, 513, 'uniform'), (50_021, 128, 600, 'popular'),
                                        (97, 256, 64, 'uniform'), (40_009, 128, 300, 'given'),
                                        (1009, 64, 40, 'uniform'), (2003, 128, 33, 'popular'), (211, 64, 1, 'uniform')])     # one-workgroup sort: items + users <= 4096 elements
def test_bpr_sgd_step_in_forward_equals_all_sorted(ra, N, d, B, kind):
    """fused.bpr_sgd_step with the in-forward update (rows touched by exactly one element of the step are rewritten by
    the wave that has them in registers; only the shared rows go through the apply pass) == the all-sorted step: same loss,
    same negatives, item and user tables equal (solo rows bit for bit; shared rows up to fp32 summation order), the
    padding row untouched, bit-reproducible; the classification == torch.bincount.  N = 97: almost every row is shared;
    'given': ids with planted collisions between positives and negatives and padding ids among the negatives."""
    n, U, lr = 64, 211, 0.3
    iw, uw = _tables(N, U, d, B)
    g = torch.Generator().manual_seed(N)
    uid = torch.randint(1, U, (B,), generator=g).to(DEV)
    pos = torch.randint(1, N, (B,), generator=g).to(DEV)
    kw = {}
    if kind == 'uniform':
        kw['sampler'] = ra.UniformSampler(N)
    elif kind == 'popular':
        counts = (torch.rand(N, generator=g) ** 6 * 1000).long()
        kw['sampler'] = ra.PopularSamplerModel(counts).to(DEV)
    else:
        neg = torch.randint(1, N, (B, n), generator=g)
        neg[:, 0] = pos.cpu().roll(1)              # a negative that is another query's positive
        neg[::3, 1] = 0                            # padding ids among the negatives
        neg[::5, 2] = neg[::5, 3]                  # the same item twice in one query
        kw['neg_ids'] = neg.to(DEV)
    results = []
    for mode in (False, True, True):
        item, user = iw.to(DEV).clone(), uw.to(DEV).clone()
        torch.manual_seed(99)
        loss, ids = ra.fused.bpr_sgd_step(item, user, n, lr, user_ids=uid, pos_ids=pos, in_forward=mode, **kw)
        results.append((loss.clone(), ids.clone(), item, user))
    (l0, i0, it0, us0), (l1, i1, it1, us1), (l2, i2, it2, us2) = results
    assert torch.equal(i0, i1) and torch.equal(l0, l1)
    assert torch.equal(it1, it2) and torch.equal(us1, us2) and torch.equal(l1, l2)          # run to run
    assert torch.equal(us0, us1)                                                               # the user side is the same code
    rel_close(it1.cpu(), it0.cpu(), rtol=1e-5, atol=1e-7)
    assert not it1[0].any() and not torch.equal(it1, iw.to(DEV))
    # the classification: solo <=> the only element of the step on its row (and not the padding row); those rows must be
    # exactly the all-sorted result
    solo_flags, _ = ra.ops.sort_step_elements(pos, i0, N)
    cnt = torch.bincount(torch.cat([pos, i0.reshape(-1)]), minlength=N)
    all_ids = torch.cat([pos.view(-1, 1), i0], 1)
    assert torch.equal(solo_flags.bool(), (cnt[all_ids] == 1) & (all_ids != 0))
    solo = (cnt == 1)
    solo[0] = False
    assert torch.equal(it1[solo], it0[solo])
    # against torch: dense SGD on the same negatives
    item_r, user_r = iw.to(DEV).clone().requires_grad_(True), uw.to(DEV).clone().requires_grad_(True)
    q = user_r[uid]
    ref = -torch.nn.functional.logsigmoid((q * item_r[pos]).sum(-1, keepdim=True) - (q.unsqueeze(1) * item_r[i0]).sum(-1)).mean(-1).mean()
    ref.backward()
    gi = item_r.grad.clone()
    gi[0] = 0
    rel_close(it1.cpu(), (iw.to(DEV) - lr * gi).cpu(), rtol=2e-4, atol=1e-6)
    rel_close(l1.cpu(), ref.detach().cpu(), rtol=1e-5)


@pytest.mark.parametrize('kind', ['uniform', 'popular'])
def test_prefetched_sgd_steps_equal_the_plain_sequence(ra, kind):
    """fused.PrefetchedBPRSGD: the next step's negatives are drawn, sorted and classified on a side stream while the
    current step's forward and apply passes run -- five steps over changing batches give the losses, negatives and tables
    of five ``bpr_sgd_step`` calls bit for bit (the same kernels on the same data, only issued earlier)."""
    N, U, d, B, n, lr, steps = 50_021, 3001, 128, 4096, 64, 0.2, 5
    iw, uw = _tables(N, U, d, B)
    g = torch.Generator().manual_seed(7)
    batches = [(torch.randint(1, U, (B,), generator=g).to(DEV), torch.randint(1, N, (B,), generator=g).to(DEV)) for _ in range(steps)]
    if kind == 'uniform':
        sampler = ra.UniformSampler(N)
    else:
        sampler = ra.PopularSamplerModel((torch.rand(N, generator=g) ** 6 * 1000).long()).to(DEV)
    item0, user0 = iw.to(DEV).clone(), uw.to(DEV).clone()
    torch.manual_seed(5)
    want = [ra.fused.bpr_sgd_step(item0, user0, n, lr, user_ids=u, pos_ids=p, sampler=sampler) for u, p in batches]
    want = [(l.clone(), i.clone()) for l, i in want]
    for _ in range(2):                                   # twice: run to run as well
        item1, user1 = iw.to(DEV).clone(), uw.to(DEV).clone()
        torch.manual_seed(5)
        stepper = ra.fused.PrefetchedBPRSGD(item1, user1, n, lr, sampler)
        got = []
        ticket = stepper.prepare(*batches[0])
        for k in range(steps):
            nxt = stepper.prepare(*batches[k + 1]) if k + 1 < steps else None
            loss, ids = stepper.step(ticket)
            got.append((loss.clone(), ids.clone()))
            ticket = nxt
        torch.cuda.synchronize()
        for (lw, iw_), (lg, ig) in zip(want, got):
            assert torch.equal(iw_, ig) and torch.equal(lw, lg)
        assert torch.equal(item1, item0) and torch.equal(user1, user0)
    assert not torch.equal(item0, iw.to(DEV))


@pytest.mark.parametrize('kind', ['uniform', 'popular'])
def test_prefetched_adam_steps_equal_the_plain_sequence(ra, kind):
    """FusedBPRAdam.prepare / step_prepared (sampling + item-side sort one batch ahead on a side stream, apply pass through
    rsa_rows_update_presorted) == the same sequence of FusedBPRAdam.step calls: losses, negatives, weights and both moment
    tables bit for bit."""
    N, U, d, B, n, steps = 50_021, 3001, 128, 4096, 64, 4
    iw, uw = _tables(N, U, d, B)
    g = torch.Generator().manual_seed(8)
    batches = [(torch.randint(1, U, (B,), generator=g).to(DEV), torch.randint(1, N, (B,), generator=g).to(DEV)) for _ in range(steps)]
    if kind == 'uniform':
        sampler = ra.UniformSampler(N)
    else:
        sampler = ra.PopularSamplerModel((torch.rand(N, generator=g) ** 6 * 1000).long()).to(DEV)
    item0, user0 = iw.to(DEV).clone(), uw.to(DEV).clone()
    torch.manual_seed(6)
    plain = ra.fused.FusedBPRAdam(item0, user0, lr=0.01)
    want = [plain.step(n, user_ids=u, pos_ids=p, sampler=sampler) for u, p in batches]
    want = [(l.clone(), i.clone()) for l, i in want]
    item1, user1 = iw.to(DEV).clone(), uw.to(DEV).clone()
    torch.manual_seed(6)
    ahead = ra.fused.FusedBPRAdam(item1, user1, lr=0.01)
    ticket = ahead.prepare(n, user_ids=batches[0][0], pos_ids=batches[0][1], sampler=sampler)
    for k in range(steps):
        nxt = ahead.prepare(n, user_ids=batches[k + 1][0], pos_ids=batches[k + 1][1], sampler=sampler) if k + 1 < steps else None
        loss, ids = ahead.step_prepared(ticket)
        assert torch.equal(ids, want[k][1]) and torch.equal(loss, want[k][0])
        ticket = nxt
    torch.cuda.synchronize()
    assert torch.equal(item1, item0) and torch.equal(user1, user0) and not torch.equal(item0, iw.to(DEV))
    for k in ('im', 'iv', 'um', 'uv'):
        assert torch.equal(ahead.state[k], plain.state[k])

