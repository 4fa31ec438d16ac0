"""GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle (oracle/), the committed reference fixtures (tests/golden) and -- as a
secondary cross-check of the random stream -- torch's own device generators."""
import numpy as np
import pytest
import torch

import oracle
from oracle import philox

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


def props(ra):
    return ra.rng.device_props(DEV)


def rel_close(a, b, rtol=1e-4, atol=1e-6):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


# --------------------------------------------------------------------------- random stream
@pytest.mark.parametrize('numel,high,seed', [(1, 1575, 2022), (1000, 1575, 2022), (4096 * 64, 10 ** 7 + 1, 1),
                                             (2048 * 256 + 77, 1000, 3), (3 * 2048 * 256 + 5, 10 ** 6, 4),
                                             (5000, 2 ** 28 + 3, 5), (2048 * 256 * 2 + 9, 2 ** 31 - 1, 6)])
def test_device_stream_matches_torch_randint(ra, numel, high, seed):
    gen = torch.cuda.default_generators[torch.cuda.current_device()]
    torch.manual_seed(seed)
    torch.rand(7, device=DEV)                       # move the offset off zero
    off0 = gen.get_offset()
    want = torch.randint(1, high, (numel,), device=DEV)
    off_torch = gen.get_offset()
    torch.manual_seed(seed)
    torch.rand(7, device=DEV)
    got = ra.ops.sample_uniform(numel, 1, high, DEV)
    assert gen.get_offset() == off_torch            # generator advanced exactly like torch does
    assert torch.equal(got, want)
    cu, mt = props(ra)
    g = philox.rng_grid_threads(numel, cu, mt)
    rest = philox.device_randint(seed, off0, numel, 1, high, g)
    assert np.array_equal(rest, want.cpu().numpy())  # oracle restatement == torch == kernel
    # G ranks drawing their row blocks of the same global call (elem_base): identical ids, identical generator advance
    for world in (2, 8):
        per = numel // world
        if per == 0:
            continue
        torch.manual_seed(seed)
        want_g = torch.randint(1, high, (per * world,), device=DEV)
        off_g = gen.get_offset()
        parts = []
        for r in range(world):
            torch.manual_seed(seed)
            with ra.rng.sharded_stream(r, world, None):
                parts.append(ra.ops.sample_uniform(per, 1, high, DEV))
            assert gen.get_offset() == off_g
        assert torch.equal(torch.cat(parts), want_g)


@pytest.mark.parametrize('numel,seed', [(3, 2022), (4096 * 64, 1), (2048 * 256 * 4 + 3, 2)])
def test_device_stream_matches_torch_rand(ra, golden, numel, seed):
    g = golden('popular')
    ps = ra.PopularSamplerModel(T(g['counts']), mode=0).to(DEV)
    gen = torch.cuda.default_generators[torch.cuda.current_device()]
    torch.manual_seed(seed)
    off0 = gen.get_offset()
    want_u = torch.rand(numel, device=DEV)
    off_torch = gen.get_offset()
    want_ids = torch.searchsorted(ps.table, want_u).clamp_(max=ps.table.numel() - 1)
    torch.manual_seed(seed)
    ids, logp, u = ra.ops.sample_popular(ps.table, ps.pop_prob, ps.guide, ps.guide_log2, numel, want_u=True)
    assert gen.get_offset() == off_torch
    # G-invariant form: G ranks drawing their slices of ONE global call (rng.sharded_stream) == the single call
    for world in (2, 8):
        per = numel // world
        if per == 0:
            continue
        parts = []
        for r in range(world):
            torch.manual_seed(seed)
            with ra.rng.sharded_stream(r, world, None):
                parts.append(ra.ops.sample_popular(ps.table, ps.pop_prob, ps.guide, ps.guide_log2, per)[0])
            assert gen.get_offset() == off0 + ra.rng.counter_offset(per * world, ra.rng.grid_threads(per * world, *props(ra)), 4)
        assert torch.equal(torch.cat(parts), want_ids_of(ps, seed, per * world))
    assert torch.equal(u, want_u)
    assert torch.equal(ids, want_ids)
    cu, mt = props(ra)
    rest = philox.device_rand(seed, off0, numel, philox.rng_grid_threads(numel, cu, mt))
    assert np.array_equal(rest, want_u.cpu().numpy())
    rel_close(logp.cpu(), torch.log(ps.pop_prob[ids]).cpu(), rtol=1e-6)


# --------------------------------------------------------------------------- popularity sampler
def want_ids_of(ps, seed, numel):
    torch.manual_seed(seed)
    return torch.searchsorted(ps.table, torch.rand(numel, device=DEV)).clamp_(max=ps.table.numel() - 1)


def test_popular_lookup_golden(ra, golden):
    """ids / log-probs for the reference's own (pop_prob, table) buffers and recorded uniforms.
    (The buffers themselves come out of torch CPU ops whose last bit depends on the host's SIMD
    width and thread count, so bit-equality of the tables is asserted on the build host --
    tests/test_host_logic.py -- and here only to 1e-6.)"""
    g = golden('popular')
    counts = T(g['counts'])
    for mode in (0, 1, 2):
        built = ra.PopularSamplerModel(counts.clone(), mode=mode)
        rel_close(built.pop_prob.numpy(), g[f'm{mode}_pop_prob'], rtol=2e-6, atol=0)
        rel_close(built.table.numpy(), g[f'm{mode}_table'], rtol=2e-6, atol=0)
        for glog in (None, 4, 9, 16, 25):      # 25: finer than fp32's 2^-24 grid (float64 cut points)
            ps = ra.PopularSamplerModel.from_tables(T(g[f'm{mode}_pop_prob']), T(g[f'm{mode}_table']), glog).to(DEV)
            want = np.minimum(g[f'm{mode}_ids'], len(counts) - 1)
            for lut in (None, ps.cdf_lut):        # binary search inside the guide bucket / direct lookup table
                ids, logp = ra.ops.popular_lookup(ps.table, ps.pop_prob, ps.guide, ps.guide_log2,
                                                  T(g[f'm{mode}_u']).to(DEV), cdf_lut=lut)
                assert np.array_equal(ids.cpu().numpy(), want)
                rel_close(logp.cpu(), g[f'm{mode}_logp'], rtol=1e-6, atol=1e-7)
    # larger table: ids from this host's build of the table == torch.searchsorted on the same table
    ps = ra.PopularSamplerModel(T(g['big_counts']), mode=0, lookup='lut')
    rel_close(ps.table[-64:].numpy(), g['big_table_tail'], rtol=2e-6, atol=0)
    u = T(g['big_u'])
    want = torch.searchsorted(ps.table, u).clamp_(max=ps.table.numel() - 1)
    ps = ps.to(DEV)
    ids, _ = ra.ops.popular_lookup(ps.table, ps.pop_prob, ps.guide, ps.guide_log2, u.to(DEV))
    assert torch.equal(ids.cpu(), want)
    # buckets holding many CDF boundaries (coarse guide): the 4-wide probe and its binary-search tail, both the
    # fused kernel's pair layout and the stand-alone kernel's separate arrays, on dense random uniforms
    coarse = ra.PopularSamplerModel.from_tables(ps.pop_prob.cpu(), ps.table.cpu(), 6, lookup='lut').to(DEV)
    torch.manual_seed(3)
    uu = torch.rand(200_000, device=DEV)
    want = torch.searchsorted(coarse.table, uu).clamp_(max=coarse.table.numel() - 1)
    ids, logp = ra.ops.popular_lookup(coarse.table, coarse.pop_prob, coarse.guide, coarse.guide_log2, uu,
                                      cdf_lut=coarse.cdf_lut)
    assert torch.equal(ids, want)
    rel_close(logp.cpu(), torch.log(coarse.pop_prob[want]).cpu(), rtol=1e-6, atol=1e-7)


def test_sampler_plugin_surface(ra):
    N = 5000
    us = ra.UniformSampler(N)
    q = torch.zeros(6, 3, 8, device=DEV)
    pos = torch.randint(1, N, (6, 3), device=DEV)
    torch.manual_seed(11)
    pp, neg, npb = us(q, 5, pos)
    torch.manual_seed(11)
    want = torch.randint(1, N, (18, 5), device=DEV).view(6, 3, 5)
    assert torch.equal(neg, want) and neg.dtype == torch.int64
    assert pp.dtype == torch.int64 and npb.dtype == torch.int64 and not pp.any() and not npb.any()
    assert pp.shape == pos.shape and npb.shape == neg.shape
    neg2, npb2 = us(7, 4, device=torch.device(DEV))
    assert neg2.shape == (7, 4) and int(neg2.min()) >= 1 and int(neg2.max()) <= N - 1
    counts = torch.randint(0, 50, (N,))
    ps = ra.PopularSamplerModel(counts).to(DEV)
    ref = oracle.PopularSamplerModel(counts)
    torch.manual_seed(12)
    pp, neg, npb = ps(q, 9, pos)
    torch.manual_seed(12)
    u = torch.rand(18, 9, device=DEV)
    want = torch.searchsorted(ps.table, u).view(6, 3, 9)
    assert torch.equal(neg, want)
    rel_close(npb.cpu(), ref.compute_item_p(neg.cpu()), rtol=1e-6)
    rel_close(pp.cpu(), ref.compute_item_p(pos.cpu()), rtol=1e-6)


# --------------------------------------------------------------------------- fused forward
def _tables(N, U, d, seed):
    g = torch.Generator().manual_seed(seed)
    iw = torch.randn(N, d, generator=g) * 0.3
    iw[0] = 0
    uw = torch.randn(U, d, generator=g) * 0.3
    uw[0] = 0
    return iw, uw


@pytest.mark.parametrize('d', [32, 64, 128, 256, 100, 512, 8])
@pytest.mark.parametrize('n', [1, 5, 64, 128, 100])
@pytest.mark.parametrize('cosine', [False, True])
def test_fused_forward_given_ids(ra, d, n, cosine):
    N, U, B = 777, 55, 37
    iw, uw = _tables(N, U, d, 100 + d + n)
    g = torch.Generator().manual_seed(d * 7 + n)
    uid = torch.randint(1, U, (B,), generator=g)
    pos = torch.randint(1, N, (B,), generator=g)
    neg = torch.randint(1, N, (B, n), generator=g)
    out = ra.ops.fused_forward(iw.to(DEV), uw.to(DEV), n, query_index=uid.to(DEV), pos_ids=pos.to(DEV),
                               neg_ids=neg.to(DEV), cosine=cosine)
    ps, ns = oracle.retriever_forward(iw, uw[uid], pos, neg, cosine=cosine)
    rel_close(out['pos_score'].cpu(), ps, atol=1e-5)
    rel_close(out['neg_score'].cpu(), ns, atol=1e-5)
    # direct query vectors instead of (user table, index)
    out2 = ra.ops.fused_forward(iw.to(DEV), uw[uid].to(DEV), n, pos_ids=pos.to(DEV), neg_ids=neg.to(DEV), cosine=cosine)
    assert torch.equal(out2['neg_score'], out['neg_score']) and torch.equal(out2['pos_score'], out['pos_score'])


@pytest.mark.parametrize('n,B', [(1, 512), (64, 100), (256, 9), (100, 33), (1024, 3)])
@pytest.mark.parametrize('kind', ['uniform', 'popular', 'popular_coarse'])
def test_fused_forward_sampled(ra, n, B, kind):
    N, U, d = 20011, 300, 128
    iw, uw = _tables(N, U, d, 5)
    g = torch.Generator().manual_seed(n + B)
    uid = torch.randint(1, U, (B,), generator=g)
    pos = torch.randint(1, N, (B,), generator=g)
    counts = (torch.rand(N, generator=g) ** 3 * 400).long()
    cu, mt = props(ra)
    gen = torch.cuda.default_generators[torch.cuda.current_device()]
    seed = 2022
    if kind == 'uniform':
        sampler = ra.UniformSampler(N)
        ref = oracle.UniformSampler(N)
    else:
        # 'popular_coarse': 2^5 guide buckets over 20 k items -> every draw takes the wide probe and its
        # binary-search tail instead of the direct LUT hit
        sampler = ra.PopularSamplerModel(counts, guide_log2=5 if kind == 'popular_coarse' else None).to(DEV)
        ref = oracle.PopularSamplerModel(counts)
        kind = 'popular'
    torch.manual_seed(seed)
    off0 = gen.get_offset()
    score, ids = ra.retriever_scores(iw.to(DEV), uw.to(DEV), n, query_index=uid.to(DEV), pos_ids=pos.to(DEV),
                                     sampler=sampler)
    gt = philox.rng_grid_threads(B * n, cu, mt)
    lpp, want_ids, lnp = ref.forward_device_stream(torch.zeros(B, 1), n, seed, off0, gt, pos_items=pos)
    assert torch.equal(ids.cpu(), want_ids)                       # bit-exact sampled negatives
    ps, ns = oracle.retriever_forward(iw, uw[uid], pos, want_ids)
    rel_close(score['pos_score'].cpu(), ps, atol=1e-5)
    rel_close(score['neg_score'].cpu(), ns, atol=1e-5)
    if kind == 'popular':
        rel_close(score['log_neg_prob'].cpu(), lnp, rtol=1e-5)
        rel_close(score['log_pos_prob'].cpu(), lpp, rtol=1e-5)
    else:
        assert score['log_neg_prob'].dtype == torch.int64 and not score['log_neg_prob'].any()
    # the stand-alone sampler plugin draws the same ids from the same generator state
    torch.manual_seed(seed)
    _, ids2, _ = sampler(torch.zeros(B, d, device=DEV), n, pos.to(DEV))
    assert torch.equal(ids2, ids)


def test_fused_forward_seq_targets_mask_padding(ra):
    # 2-D targets: pos [B,L] with padding 0 -> pos_score = -inf (baseretriever.py:164-165)
    N, d, B, L, n = 500, 64, 5, 7, 3
    iw, _ = _tables(N, 4, d, 9)
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, L, d, generator=g)
    pos = torch.randint(1, N, (B, L), generator=g)
    pos[0, 5:] = 0
    pos[3, 2:] = 0
    neg = torch.randint(1, N, (B, L, n), generator=g)
    out = ra.ops.fused_forward(iw.to(DEV), q.view(-1, d).to(DEV), n, pos_ids=pos.view(-1).to(DEV),
                               neg_ids=neg.view(-1, n).to(DEV), mask_pad_pos=True)
    ps, ns = oracle.retriever_forward(iw, q, pos, neg)
    got = out['pos_score'].cpu().view(B, L)
    assert torch.equal(torch.isinf(got), torch.isinf(ps))
    rel_close(got[~torch.isinf(ps)], ps[~torch.isinf(ps)], atol=1e-5)
    rel_close(out['neg_score'].cpu().view(B, L, n), ns, atol=1e-5)


def test_scorer_plugin_golden(ra, golden):
    g = golden('score')
    for d in (64, 128):
        for case in ('bd_bd', 'bd_bnd', 'bld_bld', 'bld_blnd'):
            k = f'd{d}_{case}'
            q, it = T(g[k + '_q']).to(DEV), T(g[k + '_items']).to(DEV)
            rel_close(ra.InnerProductScorer()(q, it).cpu(), g[k + '_ip'], atol=1e-5)
            rel_close(ra.CosineScorer()(q, it).cpu(), g[k + '_cos'], atol=1e-5)
            rel_close(ra.EuclideanScorer()(q, it).cpu(), g[k + '_euc'], atol=1e-3)
            rel_close(ra.NormScorer(p=2)(q, it).cpu(), g[k + '_norm2'], rtol=1e-4, atol=1e-4)
            if k + '_gmf' in g and case != 'bld_bld':        # the reference's GMF handles 2-D queries only
                gmf = ra.GMFScorer(d, bias=True).to(DEV)
                with torch.no_grad():
                    gmf.W.weight.copy_(T(g[k + '_gmf_w']))
                    gmf.W.bias.copy_(T(g[k + '_gmf_b']))
                rel_close(gmf(q, it).detach().cpu(), g[k + '_gmf'], rtol=1e-4, atol=1e-5)
    # GMF / Norm gradients against torch autograd of the oracle's op sequence
    torch.manual_seed(0)
    q = torch.randn(9, 64, device=DEV, requires_grad=True)
    it = torch.randn(9, 5, 64, device=DEV, requires_grad=True)
    gmf = ra.GMFScorer(64, bias=True, activation='tanh').to(DEV)
    for fn, ref in ((gmf, lambda a, b: oracle.gmf_score(a, b, gmf.W.weight, gmf.W.bias, torch.tanh)),
                    (ra.NormScorer(), oracle.norm_score)):
        grads = []
        for f in (fn, ref):
            q.grad = it.grad = None
            gmf.zero_grad()
            f(q, it).square().sum().backward()
            grads.append((q.grad.clone(), it.grad.clone(), gmf.W.weight.grad.clone() if gmf.W.weight.grad is not None else None))
        for a, b in zip(*grads):
            if a is not None:
                rel_close(a.cpu(), b.cpu(), rtol=3e-4, atol=1e-5)


# --------------------------------------------------------------------------- losses
def test_losses_golden(ra, golden):
    g = golden('loss')
    for name, cls in (('bpr_1d', 'BPRLoss'), ('bpr_2d', 'BPRLoss'), ('bpr_big', 'BPRLoss'),
                      ('ssm_1d_f32', 'SampledSoftmaxLoss'), ('ssm_1d_i64', 'SampledSoftmaxLoss'),
                      ('ssm_2d', 'SampledSoftmaxLoss'), ('ssm_big', 'SampledSoftmaxLoss'),
                      ('bce_1d', 'BinaryCrossEntropyLoss'), ('bce_2d_pad', 'BinaryCrossEntropyLoss')):
        pos = T(g[name + '_pos_score']).to(DEV).requires_grad_(True)
        neg = T(g[name + '_neg_score']).to(DEV).requires_grad_(True)
        lpp, lnp = T(g[name + '_log_pos_prob']).to(DEV), T(g[name + '_log_neg_prob']).to(DEV)
        loss = getattr(ra, cls)()(label=None, pos_score=pos, log_pos_prob=lpp, neg_score=neg, log_neg_prob=lnp)
        rel_close(loss.detach().cpu(), g[name + '_loss'], rtol=1e-5)
        (loss * 1.0).backward()
        rel_close(pos.grad.cpu(), g[name + '_grad_pos_score'], rtol=1e-4, atol=1e-7)
        rel_close(neg.grad.cpu(), g[name + '_grad_neg_score'], rtol=1e-4, atol=1e-7)


def test_other_pairwise_losses_golden(ra, golden):
    """WeightedBPR / WeightedBCE / Hinge / InfoNCE / NCE / CCL (loss_func.py:93-97, :135-193): value and gradients
    recorded from the reference."""
    g = golden('loss')
    make = {'wbpr': lambda n: ra.WeightedBPRLoss(), 'wbce': lambda n: ra.WeightedBinaryCrossEntropyLoss(),
            'hinge': lambda n: ra.HingeLoss(margin=0.5 if n == 'hinge_inactive' else 2), 'infonce': lambda n: ra.InfoNCELoss(),
            'ccl': lambda n: ra.CCLLoss(margin=0.6, neg_weight=0.3), 'nce': lambda n: ra.NCELoss()}
    for name in ('wbpr_1d', 'wbpr_2d', 'wbpr_big', 'wbce_1d', 'wbce_2d', 'wbce_2d_pad', 'hinge_1d', 'hinge_2d',
                 'hinge_inactive', 'infonce_1d', 'infonce_2d', 'ccl_1d', 'ccl_2d', 'nce_1d'):
        pos = T(g[name + '_pos_score']).to(DEV).requires_grad_(True)
        neg = T(g[name + '_neg_score']).to(DEV).requires_grad_(True)
        lpp, lnp = T(g[name + '_log_pos_prob']).to(DEV), T(g[name + '_log_neg_prob']).to(DEV)
        loss = make[name.split('_')[0]](name)(label=None, pos_score=pos, log_pos_prob=lpp, neg_score=neg, log_neg_prob=lnp)
        rel_close(loss.detach().cpu(), g[name + '_loss'], rtol=1e-5, atol=1e-7)
        (loss * 1.0).backward()
        rel_close(pos.grad.cpu(), g[name + '_grad_pos_score'], rtol=1e-4, atol=1e-7)
        rel_close(neg.grad.cpu(), g[name + '_grad_neg_score'], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('n', [1, 3, 20, 64, 300])
def test_other_pairwise_losses_vs_oracle(ra, n):
    """The same losses at other row lengths (1 / 4 / 16 / 64 lanes per row) against the oracle's autograd."""
    M = 77
    g = torch.Generator().manual_seed(n)
    pos, neg = torch.randn(M, generator=g) * 2, torch.randn(M, n, generator=g) * 2
    lpp, lnp = torch.log(torch.rand(M, generator=g)), torch.log(torch.rand(M, n, generator=g))
    pos_pad = pos.clone()
    pos_pad[5] = -float('inf')
    cases = (('wbpr', ra.WeightedBPRLoss(), lambda p, q: oracle.weighted_bpr_loss(p, q, lnp), pos),
             ('wbce', ra.WeightedBinaryCrossEntropyLoss(), lambda p, q: oracle.weighted_bce_loss(p, q, lnp), pos_pad),
             ('hinge', ra.HingeLoss(margin=1.5), lambda p, q: oracle.hinge_loss(p, q, 1.5), pos),
             ('nce', ra.NCELoss(), lambda p, q: oracle.nce_loss(p, lpp, q, lnp), pos),
             ('ccl', ra.CCLLoss(0.55, 0.4), lambda p, q: oracle.ccl_loss(p, q, 0.55, 0.4), pos))
    for name, mod, ref, pp in cases:
        p, q = pp.clone().requires_grad_(True), neg.clone().requires_grad_(True)
        want = ref(p, q)
        want.backward()
        pd, qd = pp.to(DEV).requires_grad_(True), neg.to(DEV).requires_grad_(True)
        got = mod(None, pd, lpp.to(DEV), qd, lnp.to(DEV))
        rel_close(got.detach().cpu(), want.detach(), rtol=1e-5, atol=1e-7)
        got.backward()
        rel_close(pd.grad.cpu(), torch.nan_to_num(p.grad), rtol=1e-4, atol=1e-7)
        rel_close(qd.grad.cpu(), q.grad, rtol=1e-4, atol=1e-7)


def test_ssm_shared_negatives_golden(ra, golden):
    """loss_func.py:84-89: multi-positive rows sharing one negative set, with -inf padded positives."""
    g = golden('loss')
    name = 'ssm_shared_pad'
    pos = T(g[name + '_pos_score']).to(DEV).requires_grad_(True)
    neg = T(g[name + '_neg_score']).to(DEV).requires_grad_(True)
    loss = ra.SampledSoftmaxLoss()(None, pos, T(g[name + '_log_pos_prob']).to(DEV), neg, T(g[name + '_log_neg_prob']).to(DEV))
    rel_close(loss.detach().cpu(), g[name + '_loss'], rtol=1e-5)
    loss.backward()
    rel_close(pos.grad.cpu(), g[name + '_grad_pos_score'], rtol=1e-4, atol=1e-7)
    rel_close(neg.grad.cpu(), g[name + '_grad_neg_score'], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('n', [1, 2, 7, 33, 64, 256, 1000])
@pytest.mark.parametrize('kind', ['bpr', 'ssm'])
def test_losses_vs_oracle(ra, n, kind):
    M = 129
    g = torch.Generator().manual_seed(n)
    pos, neg = torch.randn(M, generator=g) * 3, torch.randn(M, n, generator=g) * 3
    lpp, lnp = torch.log(torch.rand(M, generator=g)), torch.log(torch.rand(M, n, generator=g))
    p, q = pos.clone().requires_grad_(True), neg.clone().requires_grad_(True)
    want = oracle.bpr_loss(p, q) if kind == 'bpr' else oracle.sampled_softmax_loss(p, lpp, q, lnp)
    want.backward()
    k = ra._native.LOSS_BPR if kind == 'bpr' else ra._native.LOSS_SSM
    loss, dpos, dneg, _ = ra.ops.pairwise_loss(k, pos.to(DEV), neg.to(DEV), lpp.to(DEV), lnp.to(DEV))
    rel_close(loss.cpu(), want.detach(), rtol=1e-5)
    rel_close(dpos.cpu(), p.grad, rtol=1e-4, atol=1e-8)
    rel_close(dneg.cpu(), q.grad, rtol=1e-4, atol=1e-8)


def test_ssm_padded_positive_is_nan_like_reference(ra):
    pos = torch.tensor([0.5, -float('inf'), 1.0])
    neg = torch.randn(3, 4)
    want = oracle.sampled_softmax_loss(pos, torch.zeros(3), neg, torch.zeros(3, 4))
    loss, _, _, row = ra.ops.pairwise_loss(ra._native.LOSS_SSM, pos.to(DEV), neg.to(DEV))
    assert torch.isnan(want) and torch.isnan(loss.cpu())
    assert torch.isfinite(row.cpu()[[0, 2]]).all()


# --------------------------------------------------------------------------- backward
def test_training_step_golden(ra, golden):
    """forward -> loss -> backward through autograd == the reference's BaseRetriever.training_step +
    loss.backward() (dense weight.grad, row 0 untouched)."""
    g = golden('forward')
    for tag, lossname in (('bpr_ip', 'BPRLoss'), ('ssm_ip', 'SampledSoftmaxLoss')):
        iw = T(g[tag + '_item_w']).to(DEV).requires_grad_(True)
        uw = T(g[tag + '_user_w']).to(DEV).requires_grad_(True)
        uid, pos, neg = (T(g[tag + k]).to(DEV) for k in ('_uid', '_pos', '_neg'))
        score, ids = ra.retriever_scores(iw, uw, neg.shape[1], query_index=uid, pos_ids=pos, neg_ids=neg)
        rel_close(score['pos_score'].detach().cpu(), g[tag + '_pos_score'], atol=1e-6)
        rel_close(score['neg_score'].detach().cpu(), g[tag + '_neg_score'], atol=1e-6)
        loss = getattr(ra, lossname)()(label=None, pos_score=score['pos_score'], neg_score=score['neg_score'],
                                       log_pos_prob=T(g[tag + '_lpp']).to(DEV), log_neg_prob=T(g[tag + '_lnp']).to(DEV))
        rel_close(loss.detach().cpu(), g[tag + '_loss'], rtol=1e-5)
        loss.backward()
        rel_close(iw.grad.cpu(), g[tag + '_item_grad'], rtol=1e-4, atol=1e-7)
        rel_close(uw.grad.cpu(), g[tag + '_user_grad'], rtol=1e-4, atol=1e-7)
        assert not iw.grad[0].any()
    # cosine forward parity (scores) on the same fixture family
    tag = 'bpr_cos'
    out = ra.ops.fused_forward(T(g[tag + '_item_w']).to(DEV), T(g[tag + '_user_w']).to(DEV), g[tag + '_neg'].shape[1],
                               query_index=T(g[tag + '_uid']).to(DEV), pos_ids=T(g[tag + '_pos']).to(DEV),
                               neg_ids=T(g[tag + '_neg']).to(DEV), cosine=True)
    rel_close(out['pos_score'].cpu(), g[tag + '_pos_score'], atol=1e-6)
    rel_close(out['neg_score'].cpu(), g[tag + '_neg_score'], atol=1e-6)


@pytest.mark.parametrize('d', [64, 128, 100, 256])
@pytest.mark.parametrize('n', [1, 6, 64, 128, 320])
@pytest.mark.parametrize('loss', ['bpr', 'ssm'])
def test_backward_vs_oracle(ra, d, n, loss):
    N, U, B = 403, 61, 23
    iw, uw = _tables(N, U, d, d + n)
    g = torch.Generator().manual_seed(d * 3 + n)
    uid = torch.randint(1, U, (B,), generator=g)
    uid[3] = uid[4]                                   # repeated user in the batch
    pos = torch.randint(1, N, (B,), generator=g)
    neg = torch.randint(0, N, (B, n), generator=g)    # includes id 0 (padding: no gradient)
    neg[0, 0] = pos[0]
    lpp, lnp = torch.log(torch.rand(B, generator=g)), torch.log(torch.rand(B, n, generator=g))
    val, ps, ns, gi, gu = oracle.dense_grads(iw, uw, uid, pos, neg, loss=loss, log_pos_prob=lpp, log_neg_prob=lnp)
    for sparse in (False, True):
        iwd = iw.to(DEV).requires_grad_(True)
        uwd = uw.to(DEV).requires_grad_(True)
        score, _ = ra.retriever_scores(iwd, uwd, n, query_index=uid.to(DEV), pos_ids=pos.to(DEV), neg_ids=neg.to(DEV),
                                       sparse_grad=sparse)
        lf = ra.BPRLoss() if loss == 'bpr' else ra.SampledSoftmaxLoss()
        out = lf(label=None, pos_score=score['pos_score'], neg_score=score['neg_score'], log_pos_prob=lpp.to(DEV),
                 log_neg_prob=lnp.to(DEV))
        rel_close(out.detach().cpu(), val, rtol=1e-5)
        out.backward()
        gi_got = iwd.grad.to_dense() if sparse else iwd.grad
        gu_got = uwd.grad.to_dense() if sparse else uwd.grad
        rel_close(gi_got.cpu(), gi, rtol=2e-4, atol=1e-7)
        rel_close(gu_got.cpu(), gu, rtol=2e-4, atol=1e-7)
        assert not gi_got[0].any()


def test_backward_direct_query_grad(ra):
    # query vectors produced by an upstream encoder (SASRec): gradient flows to the [M, d] tensor
    N, d, M, n = 300, 128, 11, 64
    iw, _ = _tables(N, 4, d, 3)
    g = torch.Generator().manual_seed(8)
    q = torch.randn(M, d, generator=g)
    pos = torch.randint(1, N, (M,), generator=g)
    neg = torch.randint(1, N, (M, n), generator=g)
    qc, ic = q.clone().requires_grad_(True), iw.clone().requires_grad_(True)
    ps, ns = oracle.retriever_forward(ic, qc, pos, neg)
    oracle.bpr_loss(ps, ns).backward()
    qd, idv = q.to(DEV).requires_grad_(True), iw.to(DEV).requires_grad_(True)
    score, _ = ra.retriever_scores(idv, qd, n, pos_ids=pos.to(DEV), neg_ids=neg.to(DEV))
    ra.BPRLoss()(None, score['pos_score'], None, score['neg_score'], None).backward()
    rel_close(qd.grad.cpu(), qc.grad, rtol=2e-4, atol=1e-8)
    rel_close(idv.grad.cpu(), ic.grad, rtol=2e-4, atol=1e-8)


# --------------------------------------------------------------------------- gathers
def test_embedding_and_seg_gather(ra):
    N, d = 97, 128
    iw, _ = _tables(N, 4, d, 1)
    ids = torch.randint(0, N, (4, 9))
    assert torch.equal(ra.ops.embedding_gather(iw.to(DEV), ids.to(DEV)).cpu(), iw[ids])
    flat = torch.randint(1, N, (200,))
    start = torch.tensor([0, 5, 5, 40, 199, 120])
    end = torch.tensor([5, 5, 31, 41, 200, 200])        # empty, short, longer than max_len
    L = 20
    got_ids, got_rows, got_len = ra.ops.seg_gather(iw.to(DEV), flat.to(DEV), start.to(DEV), end.to(DEV), L)
    # segments longer than max_len keep the most recent L items
    s2 = torch.maximum(start, end - L)
    want_ids, want_rows, want_len = oracle.seq_gather(iw, flat, s2.tolist(), end.tolist(), L)
    assert torch.equal(got_ids.cpu(), want_ids) and torch.equal(got_len.cpu(), want_len)
    assert torch.equal(got_rows.cpu(), want_rows)
    src = torch.randn(50, d)
    idx = torch.randint(0, 9, (50,))
    want = torch.zeros(9, d).index_add_(0, idx, src)
    want[0] = 0
    rel_close(ra.ops.scatter_add_rows(src.to(DEV), idx.to(DEV), 9).cpu(), want, rtol=1e-5, atol=1e-6)


def test_errors_are_loud(ra):
    iw = torch.zeros(10, 6, device=DEV)            # dim not a multiple of 4
    with pytest.raises(ra._native.NativeError):
        ra.ops.fused_forward(iw, torch.zeros(2, 6, device=DEV), 1, neg_ids=torch.ones(2, 1, dtype=torch.int64, device=DEV))
    with pytest.raises(RuntimeError):
        ra.ops.embedding_gather(torch.zeros(4, 8), torch.zeros(2, dtype=torch.int64))   # CPU tensors: no fallback
    with pytest.raises(TypeError):
        ra.ops.embedding_gather(torch.zeros(4, 8, device=DEV), torch.zeros(2, dtype=torch.int32, device=DEV))


# --------------------------------------------------------------------------- BASELINE.json full sizes
def test_full_size_properties(ra):
    """configs[1] shape (N = 1e7 items, d = 128, B = 4096, n = 64, popularity sampler): size-independent
    properties + an oracle spot check on a random subset of elements."""
    N, U, d, B, n = 10_000_001, 1_000_001, 128, 4096, 64
    g = torch.Generator(device=DEV).manual_seed(1)
    iw = torch.empty(N, d, device=DEV).normal_(0, 0.02, generator=g)
    iw[0] = 0
    uw = torch.empty(U, d, device=DEV).normal_(0, 0.02, generator=g)
    counts = (torch.rand(N, generator=torch.Generator().manual_seed(2)) ** 8 * 1e4).long()
    ps = ra.PopularSamplerModel(counts).to(DEV)
    uid = torch.randint(1, U, (B,), device=DEV, generator=g)
    pos = torch.randint(1, N, (B,), device=DEV, generator=g)
    torch.manual_seed(2022)
    score, ids = ra.retriever_scores(iw, uw, n, query_index=uid, pos_ids=pos, sampler=ps)
    # (1) bit-exact ids vs the reference's own op sequence on this device
    torch.manual_seed(2022)
    want_ids = torch.searchsorted(ps.table, torch.rand(B, n, device=DEV)).clamp_(max=N - 1)
    assert torch.equal(ids, want_ids)
    assert int(ids.min()) >= 0 and int(ids.max()) < N
    # (2) idempotence: re-scoring the sampled ids as given ids reproduces the scores bit for bit
    again = ra.ops.fused_forward(iw, uw, n, query_index=uid, pos_ids=pos, neg_ids=ids)
    assert torch.equal(again['neg_score'], score['neg_score']) and torch.equal(again['pos_score'], score['pos_score'])
    # (3) linearity in the query
    twice = ra.ops.fused_forward(iw, uw * 2, n, query_index=uid, pos_ids=pos, neg_ids=ids)
    assert torch.equal(twice['neg_score'], score['neg_score'] * 2)
    # (4) oracle spot check on 2000 random (b, j) elements
    sel_b = torch.randint(0, B, (2000,))
    sel_j = torch.randint(0, n, (2000,))
    rows = iw[ids[sel_b.to(DEV), sel_j.to(DEV)]].cpu()
    q = uw[uid[sel_b.to(DEV)]].cpu()
    rel_close(score['neg_score'].cpu()[sel_b, sel_j], oracle.inner_product_score(q, rows), rtol=1e-4, atol=1e-7)
    rel_close(score['log_neg_prob'].cpu()[sel_b, sel_j], torch.log(ps.pop_prob.cpu()[ids.cpu()[sel_b, sel_j]]), rtol=1e-5)
    # (5) uniform sampler at the same size
    us = ra.UniformSampler(N)
    torch.manual_seed(7)
    s2, ids2 = ra.retriever_scores(iw, uw, n, query_index=uid, pos_ids=pos, sampler=us)
    torch.manual_seed(7)
    assert torch.equal(ids2, torch.randint(1, N, (B, n), device=DEV))
    # (6) loss + row-sparse backward: gradient rows sum to the analytic total
    loss, dpos, dneg, _ = ra.ops.pairwise_loss(ra._native.LOSS_BPR, score['pos_score'], score['neg_score'])
    _, rows_g, qg = ra.ops.fused_backward(iw, uw, ids, dneg, query_index=uid, pos_ids=pos, dpos=dpos,
                                          dense_item_grad=False, row_item_grad=True)
    assert torch.isfinite(loss) and rows_g.shape == (B * (n + 1), d)
    # sum_j dneg_j + dpos == 0 for BPR, so each query's gradient rows sum to ~0 * q
    tot = rows_g.view(B, n + 1, d).sum(1)
    keep = (ids != 0).all(1)                      # a sampled padding id contributes no gradient row
    assert float(tot[keep].abs().max()) < 1e-6


@pytest.mark.parametrize('n', [64, 128, 320])
@pytest.mark.parametrize('kind', ['given', 'uniform', 'popular'])
def test_fused_bpr_epilogue_matches_separate_loss_and_oracle(ra, n, kind):
    """One-launch forward+BPR (value and gradient) == separate kernels == oracle autograd."""
    from recstudio_amd.fused import fused_bpr_loss
    N, U, d, B = 2003, 97, 128, 41
    iw, uw = _tables(N, U, d, n)
    g = torch.Generator().manual_seed(n)
    uid = torch.randint(1, U, (B,), generator=g)
    pos = torch.randint(1, N, (B,), generator=g)
    counts = (torch.rand(N, generator=g) ** 3 * 100).long()
    sampler = {'given': None, 'uniform': ra.UniformSampler(N), 'popular': ra.PopularSamplerModel(counts).to(DEV)}[kind]
    neg_given = torch.randint(0, N, (B, n), generator=g).to(DEV) if kind == 'given' else None
    for sparse in (False, True):
        iwd, uwd = iw.to(DEV).requires_grad_(True), uw.to(DEV).requires_grad_(True)
        torch.manual_seed(77)
        loss, ids = fused_bpr_loss(iwd, uwd, n, query_index=uid.to(DEV), pos_ids=pos.to(DEV), sampler=sampler,
                                   neg_ids=neg_given, sparse_grad=sparse)
        (loss * 3.0).backward()                       # non-trivial upstream gradient
        val, ps, ns, gi, gu = oracle.dense_grads(iw, uw, uid, pos, ids.cpu(), loss='bpr')
        rel_close(loss.detach().cpu(), val, rtol=1e-5)
        gi_got = iwd.grad.to_dense() if sparse else iwd.grad
        gu_got = uwd.grad.to_dense() if sparse else uwd.grad
        rel_close(gi_got.cpu(), gi * 3, rtol=2e-4, atol=1e-7)
        rel_close(gu_got.cpu(), gu * 3, rtol=2e-4, atol=1e-7)
        # the sampler stream is the same as the unfused path's
        torch.manual_seed(77)
        _, ids2 = ra.retriever_scores(iw.to(DEV), uw.to(DEV), n, query_index=uid.to(DEV), pos_ids=pos.to(DEV),
                                      sampler=sampler, neg_ids=neg_given)
        assert torch.equal(ids, ids2)


@pytest.mark.parametrize('d,n', [(64, 5), (128, 64), (100, 3)])
def test_cosine_backward_vs_oracle(ra, d, n):
    N, U, B = 301, 40, 17
    iw, uw = _tables(N, U, d, d + n)
    iw[0] = 0
    g = torch.Generator().manual_seed(n)
    uid = torch.randint(1, U, (B,), generator=g)
    pos = torch.randint(1, N, (B,), generator=g)
    neg = torch.randint(1, N, (B, n), generator=g)        # no padding ids: the reference yields NaN for them
    val, ps, ns, gi, gu = oracle.dense_grads(iw, uw, uid, pos, neg, loss='bpr', cosine=True)
    iwd, uwd = iw.to(DEV).requires_grad_(True), uw.to(DEV).requires_grad_(True)
    score, _ = ra.retriever_scores(iwd, uwd, n, query_index=uid.to(DEV), pos_ids=pos.to(DEV), neg_ids=neg.to(DEV),
                                   cosine=True)
    loss = ra.BPRLoss()(None, score['pos_score'], None, score['neg_score'], None)
    rel_close(loss.detach().cpu(), val, rtol=1e-5)
    loss.backward()
    rel_close(iwd.grad.cpu(), gi, rtol=3e-4, atol=1e-7)
    rel_close(uwd.grad.cpu(), gu, rtol=3e-4, atol=1e-7)
    # the scorer plugin on materialised vectors is differentiable too (both scorers)
    for cls, ofn in ((ra.CosineScorer, oracle.cosine_score), (ra.InnerProductScorer, oracle.inner_product_score),
                     (ra.EuclideanScorer, oracle.euclidean_score)):
        q = uw[uid].clone().requires_grad_(True)
        it = iw[neg].clone().requires_grad_(True)
        ofn(q, it).square().sum().backward()
        qd = uw[uid].to(DEV).requires_grad_(True)
        itd = iw[neg].to(DEV).requires_grad_(True)
        cls()(qd, itd).square().sum().backward()
        # atol: gradient entries are sums of O(1e2) cancelling terms; allow 1e-6 of the largest entry
        rel_close(qd.grad.cpu(), q.grad, rtol=3e-4, atol=1e-6 + 1e-6 * float(q.grad.abs().max()))
        rel_close(itd.grad.cpu(), it.grad, rtol=3e-4, atol=1e-6 + 1e-6 * float(it.grad.abs().max()))


def test_masked_uniform_sampler(ra, golden):
    """MaskedUniformSampler (sampler.py:117-147): same ids as the reference's op sequence on this device's
    random stream, and the recorded reference outputs for recorded uniforms (through the oracle)."""
    g = golden('uniform')
    N, hist = int(g['mask_N']), T(g['mask_hist'])
    cu, mt = props(ra)
    gen = torch.cuda.default_generators[torch.cuda.current_device()]
    ms = ra.MaskedUniformSampler(N)
    for qdim, n in ((2, 40), (3, 5)):
        query = torch.zeros(hist.shape[0], 4, device=DEV) if qdim == 2 else torch.zeros(hist.shape[0], 3, 4, device=DEV)
        per_row = n if qdim == 2 else 3 * n
        torch.manual_seed(123)
        off0 = gen.get_offset()
        pp, neg, npb = ms(query, n, torch.ones(hist.shape[0], dtype=torch.int64, device=DEV), hist.to(DEV))
        torch.manual_seed(123)
        u = torch.rand(hist.shape[0], per_row, device=DEV)            # what the reference would draw here
        want = oracle.masked_uniform_from_u(N - 1, n, hist, u.cpu(), None if qdim == 2 else 3)
        assert torch.equal(neg.cpu(), want)
        rest = philox.device_rand(123, off0, hist.shape[0] * per_row, philox.rng_grid_threads(hist.shape[0] * per_row, cu, mt))
        assert np.array_equal(rest.reshape(u.shape), u.cpu().numpy())
        assert npb.dtype == torch.int64 and not npb.any() and npb.shape == neg.shape
    # bigger, random histories: never a history item, always in [1, N-1], roughly uniform
    N2, B2, L2, n2 = 500, 64, 300, 2000
    gg = torch.Generator().manual_seed(4)
    h2 = torch.zeros(B2, L2, dtype=torch.int64)
    for b in range(B2):
        m = int(torch.randint(0, L2 + 1, (1,), generator=gg))
        h2[b, torch.randperm(L2, generator=gg)[:m]] = torch.randperm(N2 - 1, generator=gg)[:m] + 1
    neg = ra.ops.sample_masked_uniform(h2.to(DEV), N2 - 1, n2).cpu()
    for b in range(B2):
        assert not set(neg[b].tolist()) & set(h2[b].tolist())
    assert int(neg.min()) >= 1 and int(neg.max()) <= N2 - 1


@pytest.mark.parametrize('d,n,sampler', [(128, 64, 'uniform'), (64, 128, 'popular'), (32, 64, 'given'), (256, 64, 'uniform')])
def test_forward_query_grad_matches_backward_kernel(ra, d, n, sampler):
    """rsa_fused_args.query_grad (accumulated in the forward while the rows are in registers) == the query
    gradient of rsa_fused_backward, and the backward without query outputs still writes the same item rows."""
    torch.manual_seed(4)
    N, U, M = 5003, 301, 97
    item = torch.randn(N, d, device=DEV) * 0.3
    item[0] = 0
    user = torch.randn(U, d, device=DEV) * 0.3
    uid = torch.randint(1, U, (M,), device=DEV)
    pos = torch.randint(1, N, (M,), device=DEV)
    kw = {}
    if sampler == 'given':
        kw = dict(neg_ids=torch.randint(0, N, (M, n), device=DEV))
    elif sampler == 'uniform':
        kw = dict(sampler=ra._native.SAMPLER_UNIFORM)
    else:
        ps = ra.PopularSamplerModel((torch.rand(N) ** 3 * 50).long()).to(DEV)
        kw = dict(sampler=ra._native.SAMPLER_POPULAR, table=ps.table, pop_prob=ps.pop_prob, guide=ps.guide,
                  guide_log2=ps.guide_log2, table_prob=ps.table_prob, cdf_lut=ps.cdf_lut)
    torch.manual_seed(9)
    a = ra.ops.fused_forward(item, user, n, query_index=uid, pos_ids=pos, fused_bpr=True, want_query_grad=True, **kw)
    torch.manual_seed(9)
    b = ra.ops.fused_forward(item, user, n, query_index=uid, pos_ids=pos, fused_bpr=True, **kw)
    assert torch.equal(a['neg_ids'], b['neg_ids'])
    rel_close(a['neg_score'].cpu(), b['neg_score'].cpu(), rtol=1e-5, atol=1e-6)
    rel_close(a['dneg'].cpu(), b['dneg'].cpu(), rtol=1e-4, atol=1e-9)
    rel_close(a['loss'].cpu(), b['loss'].cpu(), rtol=1e-6)
    _, rows_ref, qg_ref = ra.ops.fused_backward(item, user, b['neg_ids'], b['dneg'], query_index=uid, pos_ids=pos,
                                                dpos=b['dpos'], dense_item_grad=False, row_item_grad=True)
    rel_close(a['query_grad'].cpu(), qg_ref.cpu(), rtol=2e-4, atol=1e-8)
    ig, rows, qg = ra.ops.fused_backward(item, user, a['neg_ids'], a['dneg'], query_index=uid, pos_ids=pos,
                                         dpos=a['dpos'], dense_item_grad=True, row_item_grad=True, want_query_grad=False)
    assert qg is None
    rel_close(rows.cpu(), rows_ref.cpu(), rtol=1e-4, atol=1e-9)
    ig_ref = ra.ops.fused_backward(item, user, b['neg_ids'], b['dneg'], query_index=uid, pos_ids=pos, dpos=b['dpos'])[0]
    rel_close(ig.cpu(), ig_ref.cpu(), rtol=2e-4, atol=1e-8)
    # through autograd: the fused BPR loss's gradients against torch autograd of the same loss
    iw = item.clone().requires_grad_(True)
    uw = user.clone().requires_grad_(True)
    torch.manual_seed(9)
    loss, neg = ra.fused.fused_bpr_loss(iw, uw, n, query_index=uid, pos_ids=pos,
                                        sampler=None if sampler == 'given' else (ra.UniformSampler(N) if sampler == 'uniform' else ps),
                                        neg_ids=kw.get('neg_ids'))
    loss.backward()
    iw2 = item.clone().requires_grad_(True)
    uw2 = user.clone().requires_grad_(True)
    q = uw2[uid]
    ref = -torch.nn.functional.logsigmoid((q * iw2[pos]).sum(-1, keepdim=True) - (q.unsqueeze(1) * iw2[neg]).sum(-1)).mean(-1).mean()
    ref.backward()
    rel_close(loss.detach().cpu(), ref.detach().cpu(), rtol=1e-5)
    rel_close(uw.grad.cpu(), uw2.grad.cpu(), rtol=3e-4, atol=1e-8)
    g2 = iw2.grad.clone()
    g2[0] = 0
    rel_close(iw.grad.cpu(), g2.cpu(), rtol=3e-4, atol=1e-8)


def test_empty_inputs_are_no_ops(ra):
    """Zero queries / zero ids: every entry point returns cleanly with empty outputs (the reference's torch ops do the
    same on empty tensors)."""
    N, d = 101, 64
    item = torch.randn(N, d, device=DEV)
    e64 = torch.empty(0, dtype=torch.int64, device=DEV)
    assert ra.ops.sample_uniform(0, 1, N, DEV).numel() == 0
    assert ra.ops.embedding_gather(item, e64).shape == (0, d)
    out = ra.ops.fused_forward(item, torch.empty(0, d, device=DEV), 4, pos_ids=e64, neg_ids=torch.empty(0, 4, dtype=torch.int64, device=DEV))
    assert out['neg_score'].shape == (0, 4) and out['pos_score'].shape == (0,)
    neg, lp = ra.UniformSampler(N)(torch.empty(0, d, device=DEV), 5)
    assert neg.shape == (0, 5)
    _, lse, tv, ti = ra.ops.fullscore(item, torch.empty(0, d, device=DEV), want_lse=True, k=3)
    assert lse.shape == (0,) and tv.shape == (0, 3)
    res = ra.ops.seg_gather(item, torch.randint(1, N, (10,), device=DEV), e64, e64, 7)
    assert sorted(tuple(t.shape) for t in res) == [(0,), (0, 7), (0, 7, d)]
    g = ra.ops.scatter_rows_sorted(torch.zeros(N, d, device=DEV), torch.empty(0, d, device=DEV),
                                   torch.empty(0, 4, dtype=torch.int64, device=DEV), torch.empty(0, 4, device=DEV))
    assert not g.any()


def test_fused_forward_fuzz_against_device_reference_ops(ra):
    """40 random configurations (dim incl. non-power-of-two / generic dims, 1..300 negatives, ragged batch sizes, all
    three samplers, all three scorers): ids == the reference's torch.randint / searchsorted(rand) on this device under
    the same seed, scores == F.embedding + the scorer's formula in float64."""
    rs = np.random.RandomState(1234)
    gen = torch.cuda.default_generators[torch.cuda.current_device()]
    for it in range(40):
        d = int(rs.choice([32, 64, 100, 128, 256, 36, 512]))
        n = int(rs.choice([1, 2, 5, 33, 64, 100, 128, 300]))
        B = int(rs.randint(1, 150))
        N = int(rs.choice([50, 1000, 20011]))
        mode = int(rs.randint(0, 3))
        kind = ['given', 'uniform', 'popular'][int(rs.randint(0, 3))]
        torch.manual_seed(it)
        item = torch.randn(N, d, device=DEV) * 0.3
        q = torch.randn(B, d, device=DEV) * 0.3
        pos = torch.randint(0, N, (B,), device=DEV)
        seed = 1000 + it
        kw = {}
        if kind == 'given':
            want_ids = torch.randint(0, N, (B, n), device=DEV)
            kw['neg_ids'] = want_ids
        elif kind == 'uniform':
            kw['sampler'] = ra._native.SAMPLER_UNIFORM
            torch.manual_seed(seed)
            want_ids = torch.randint(1, N, (B, n), device=DEV)
        else:
            ps = ra.PopularSamplerModel((torch.rand(N) ** 3 * 60).long()).to(DEV)
            kw.update(sampler=ra._native.SAMPLER_POPULAR, table=ps.table, pop_prob=ps.pop_prob, guide=ps.guide,
                      guide_log2=ps.guide_log2, table_prob=ps.table_prob, cdf_lut=ps.cdf_lut)
            torch.manual_seed(seed)
            want_ids = torch.searchsorted(ps.table, torch.rand(B, n, device=DEV)).clamp_(max=N - 1)
        torch.manual_seed(seed)
        out = ra.ops.fused_forward(item, q, n, pos_ids=pos, cosine=mode, **kw)
        tag = f'it={it} d={d} n={n} B={B} N={N} mode={mode} {kind}'
        assert torch.equal(out['neg_ids'], want_ids), tag
        qd, xd, pd = q.double(), item.double()[want_ids], item.double()[pos]
        dot_n, dot_p = (qd.unsqueeze(1) * xd).sum(-1), (qd * pd).sum(-1)
        if mode == 0:
            wn, wp = dot_n, dot_p
        elif mode == 1:
            wn = dot_n / xd.norm(dim=-1) / qd.norm(dim=-1, keepdim=True)
            wp = dot_p / pd.norm(dim=-1) / qd.norm(dim=-1)
        else:
            wn = -(xd.square().sum(-1) + qd.square().sum(-1, keepdim=True) - 2 * dot_n)
            wp = -(pd.square().sum(-1) + qd.square().sum(-1) - 2 * dot_p)
        ok = torch.isfinite(wn)
        np.testing.assert_allclose(out['neg_score'][ok].cpu(), wn[ok].float().cpu(), rtol=2e-4, atol=2e-5, err_msg=tag)
        okp = torch.isfinite(wp)
        np.testing.assert_allclose(out['pos_score'][okp].cpu(), wp[okp].float().cpu(), rtol=2e-4, atol=2e-5, err_msg=tag)


def test_backward_fuzz_against_torch_autograd(ra):
    """24 random configurations: gradients of a random scalar functional of (pos_score, neg_score) through the fused
    score node (dense item gradient, user-table gradient; the sorted / atomic / cosine / Euclidean backward paths)
    == torch autograd of the same functional over F.embedding + the scorer's formula."""
    rs = np.random.RandomState(99)
    for it in range(24):
        d = int(rs.choice([32, 64, 128, 256, 48]))
        n = int(rs.choice([1, 3, 64, 128, 70]))
        B = int(rs.randint(1, 90))
        N, U = int(rs.choice([40, 3001])), 57
        mode = int(rs.randint(0, 3))
        torch.manual_seed(300 + it)
        iw = (torch.randn(N, d, device=DEV) * 0.4)
        iw[0] = 0
        uw = torch.randn(U, d, device=DEV) * 0.4
        uid = torch.randint(1, U, (B,), device=DEV)
        pos = torch.randint(1, N, (B,), device=DEV)
        neg = torch.randint(0, N, (B, n), device=DEV)
        cp, cn = torch.randn(B, device=DEV), torch.randn(B, n, device=DEV)
        a, b = iw.clone().requires_grad_(True), uw.clone().requires_grad_(True)
        score, _ = ra.retriever_scores(a, b, n, query_index=uid, pos_ids=pos, neg_ids=neg, cosine=mode)
        ((score['pos_score'] * cp).sum() + (score['neg_score'] * cn).sum()).backward()
        a2, b2 = iw.clone().double().requires_grad_(True), uw.clone().double().requires_grad_(True)
        q, xp, xn = b2[uid], a2[pos], a2[neg]
        dp, dn = (q * xp).sum(-1), (q.unsqueeze(1) * xn).sum(-1)
        if mode == 1:
            dp = dp / xp.norm(dim=-1) / q.norm(dim=-1)
            dn = dn / xn.norm(dim=-1) / q.norm(dim=-1, keepdim=True)
        elif mode == 2:
            dp = -(xp.square().sum(-1) + q.square().sum(-1) - 2 * dp)
            dn = -(xn.square().sum(-1) + q.square().sum(-1, keepdim=True) - 2 * dn)
        live = neg != 0 if mode == 1 else torch.ones_like(neg, dtype=torch.bool)      # cosine of the zero padding row: NaN
        ((dp * cp.double()).sum() + (dn * cn.double())[live].sum()).backward()
        tag = f'it={it} d={d} n={n} B={B} N={N} mode={mode}'
        want_i = a2.grad.clone()
        want_i[0] = 0                                             # the padding row never receives gradient
        if mode == 1 and not bool(live.all()):
            continue                                              # reference semantics there are NaN: not compared
        scale = float(want_i.abs().max()) + 1e-6
        np.testing.assert_allclose(a.grad.cpu(), want_i.float().cpu(), rtol=3e-4, atol=3e-6 * scale, err_msg=tag)
        want_u = b2.grad.clone()
        np.testing.assert_allclose(b.grad.cpu(), want_u.float().cpu(), rtol=3e-4, atol=3e-6 * (float(want_u.abs().max()) + 1e-6),
                                   err_msg=tag)
