"""Round 5, GPU: the batch queue of the fused forward, the two-call SGD step, evaluation without host round trips."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def ra():
    import recstudio_amd as ra
    ra._native.lib()
    return ra


@pytest.mark.parametrize('kind', ['uniform', 'popular'])
@pytest.mark.parametrize('n', [64, 128])
def test_batch_queue_equals_consecutive_launches(ra, kind, n):
    """rsa_fused_args.n_batches: S independent batches consumed by ONE launch == S consecutive launches -- the negatives of
    every batch are those of its own torch call (ids bit for bit, generator advanced S times), scores, row losses and
    d loss/d score bit-equal (the same arithmetic on the same rows), per-batch mean loss within float rounding."""
    nat = ra._native
    N, U, d, B, S = 200_003, 5001, 128, 1024, 5
    g = torch.Generator(device=DEV).manual_seed(11)
    item = torch.randn(N, d, device=DEV, generator=g) * 0.1
    user = torch.randn(U, d, device=DEV, generator=g) * 0.1
    uid = torch.randint(1, U, (S, B), device=DEV, generator=g)
    pos = torch.randint(1, N, (S, B), device=DEV, generator=g)
    kw = {'sampler': nat.SAMPLER_UNIFORM}
    if kind == 'popular':
        sampler = ra.PopularSamplerModel((torch.rand(N) ** 5 * 500).long()).to(DEV)
        kw = dict(sampler.lookup_kwargs(), sampler=nat.SAMPLER_POPULAR)
    bpr = n == 64
    torch.manual_seed(3)
    want = [ra.ops.fused_forward(item, user, n, query_index=uid[k], pos_ids=pos[k], fused_bpr=bpr, **kw) for k in range(S)]
    state_after = torch.cuda.get_rng_state(DEV).clone()
    torch.manual_seed(3)
    got = ra.ops.fused_forward(item, user, n, query_index=uid.reshape(-1), pos_ids=pos.reshape(-1), fused_bpr=bpr, n_batches=S, **kw)
    assert torch.equal(torch.cuda.get_rng_state(DEV), state_after)
    keys = ['neg_ids', 'neg_score', 'pos_score'] + (['row_loss', 'dneg', 'dpos'] if bpr else [])
    if kind == 'popular':
        keys += ['neg_logp', 'pos_logp']
    for key in keys:
        cat = torch.cat([w[key] for w in want])
        assert torch.equal(cat.reshape(-1), got[key].reshape(-1)), key
    if bpr:
        assert got['loss'].shape == (S,)
        torch.testing.assert_close(got['loss'], torch.stack([w['loss'] for w in want]), rtol=1e-6, atol=1e-7)
    # the frozen form advances the generator by S calls as well
    torch.manual_seed(3)
    st = ra.ops.FusedStep(item, user, n, query_index=uid.reshape(-1), pos_ids=pos.reshape(-1), fused_bpr=bpr, n_batches=S, **kw)
    first = st.out['neg_ids'].clone()
    st()
    assert torch.equal(first.reshape(-1), got['neg_ids'].reshape(-1)) and not torch.equal(st.out['neg_ids'], first)
    if bpr:       # the per-batch losses follow every replay (one persistent [S] buffer, recomputed from the new row losses)
        loss_buf = st.out['loss']
        torch.testing.assert_close(loss_buf, st.out['row_loss'].view(S, -1).mean(1), rtol=1e-6, atol=1e-7)
        before = loss_buf.clone()
        assert st()['loss'] is loss_buf and not torch.equal(loss_buf, before)
        torch.testing.assert_close(loss_buf, st.out['row_loss'].view(S, -1).mean(1), rtol=1e-6, atol=1e-7)


def test_batch_queue_argument_checks(ra):
    nat = ra._native
    item, user = torch.zeros(100, 64, device=DEV), torch.zeros(10, 64, device=DEV)
    uid, pos = torch.ones(6, dtype=torch.int64, device=DEV), torch.ones(6, dtype=torch.int64, device=DEV)
    with pytest.raises(ValueError, match='divide'):
        ra.ops.fused_forward(item, user, 64, query_index=uid, pos_ids=pos, sampler=nat.SAMPLER_UNIFORM, n_batches=4)
    with pytest.raises((ValueError, nat.NativeError)):
        ra.ops.fused_forward(item, user, 64, query_index=uid, pos_ids=pos, neg_ids=torch.ones(6, 64, dtype=torch.int64, device=DEV),
                             n_batches=2)


def test_eval_epoch_matches_per_batch_host_accumulation(ra):
    """_eval_epoch keeps the per-batch metric values on the device and replays cached device batches: the epoch's metric
    dict is what the reference's loop gives -- sum over batches of float(value) * batch size / total, in batch order --
    and a second epoch (served from the cache) returns the same numbers."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_dataset_golden import make
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'data_ml100k.npz'))
    model = ra.BPR({'eval': {'batch_size': 64}, 'train': {'epochs': 1}})
    ds = make(ra.TripletDataset, g)
    trn, val, tst = ds.build(split_ratio=[0.8, 0.1, 0.1], shuffle=True, split_mode='user_entry')
    model.fit(trn, val)
    dev = next(model.parameters()).device
    first = model._eval_epoch(val, model.validation_step, dev)
    again = model._eval_epoch(val, model.validation_step, dev)
    assert first == again and model._eval_cache
    model.eval()
    model._update_item_vector()
    acc, total = {}, 0
    with torch.no_grad():
        for batch in val.eval_loader(batch_size=64):
            metrics, bs = model.validation_step(model._to_device(batch, dev))
            for k, v in metrics.items():
                acc[k] = acc.get(k, 0.0) + float(v) * bs
            total += bs
    want = {k: v / total for k, v in acc.items()}
    assert first.keys() == want.keys()
    for k in want:
        assert first[k] == pytest.approx(want[k], rel=1e-12, abs=0), k


@pytest.mark.parametrize('B,n,d', [(2048, 50_000, 128), (300, 4097, 128), (33, 130, 64), (64, 1000, 32), (17, 257, 48)])
def test_probs_t_query_vs_float64(ra, B, n, d):
    """rsa_probs_t_query (the d/d items GEMM of the full-softmax backward, item-stationary fp32 MFMA) == probs.T @ query in
    float64, incl. batch sizes that are not multiples of the 32-row chunk, catalogs that end inside a 128-item block and an
    embed_dim that is zero-padded to the kernel's; fp32 accumulation over B terms: rtol 2e-5 of the largest entry."""
    g = torch.Generator(device=DEV).manual_seed(B + n)
    probs = torch.rand(B, n, device=DEV, generator=g) / n
    query = torch.randn(B, d, device=DEV, generator=g)
    got = ra.ops.probs_t_query(probs, query)
    want = probs.double().t() @ query.double()
    assert got.shape == (n, d)
    err = (got.double() - want).abs().max().item()
    assert err <= 2e-5 * want.abs().max().item(), err
    out = torch.full((n + 1, d), 7.0, device=DEV)
    ra.ops.probs_t_query(probs, query, out=out[1:])
    assert torch.equal(out[1:], got) and bool((out[0] == 7.0).all())


def test_full_softmax_backward_has_no_library_gemm(ra):
    """The full-softmax training path end to end: loss.backward() through scorer.full_lse gives the float64 gradients and runs
    no rocBLAS / hipBLASLt kernel (profiler kernel names)."""
    from recstudio_amd.scorer import full_lse
    N, B, d = 30_001, 256, 128
    g = torch.Generator(device=DEV).manual_seed(9)
    w = (torch.randn(N, d, device=DEV, generator=g) * 0.1).requires_grad_(True)
    q = (torch.randn(B, d, device=DEV, generator=g) * 0.3).requires_grad_(True)
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        loss = full_lse(q, w).mean()
        loss.backward()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    assert not [k for k in names if 'Cijk' in k or 'gemm' in k.lower()], names
    wd, qd = w.detach().double(), q.detach().double()
    wd.requires_grad_(True)
    qd.requires_grad_(True)
    torch.logsumexp(qd @ wd[1:].t(), -1).mean().backward()
    torch.testing.assert_close(w.grad.double(), wd.grad, rtol=2e-4, atol=1e-9)
    torch.testing.assert_close(q.grad.double(), qd.grad, rtol=2e-4, atol=1e-9)
    assert bool((w.grad[0] == 0).all())


def test_placement_changes_addresses_only(ra, monkeypatch):
    """recstudio_amd.placement (opt-in since round 6: RSA_PLACEMENT=1 / ``placement.enabled()``; off, every arena is a plain torch
    allocation and nothing is probed or held): switched on, output arenas of >= MIN_BYTES are probed (rsa_placement_probe) and
    taken from the fast class of allocations; verdicts are cached by address; smaller arenas and ``placement.disabled()`` are
    plain torch allocations; the results of a launch do not depend on any of it; ``release`` gives everything back."""
    from recstudio_amd import placement
    nat = ra._native
    dev = torch.device(DEV)
    monkeypatch.setattr(placement, 'SOURCE_BYTES', 320 << 20)
    monkeypatch.setattr(placement, 'SPACER_BYTES', 64 << 20)
    monkeypatch.setattr(placement, 'WARM_S', 0.02)
    monkeypatch.setattr(placement, 'ENABLED', False)
    before = placement.summary(dev)['probes']
    off = placement.pick(48 << 20, dev)                        # the default: plain, no probe, nothing resident
    assert off.numel() == 48 << 20 and placement.summary(dev)['probes'] == before and placement._st(dev)['source'] is None
    monkeypatch.setattr(placement, 'ENABLED', True)
    buf = placement.pick(48 << 20, dev)
    assert buf.dtype == torch.uint8 and buf.numel() == 48 << 20 and buf.device == dev
    after = placement.summary(dev)
    assert 1 <= after['probes'] - before <= placement.MAX_TRIES and after['picked'] >= 1
    us = placement.probe_us(buf)
    assert 5 < us < 5000
    again = placement.summary(dev)['probes']
    with placement.disabled():
        plain = placement.pick(48 << 20, dev)
    assert placement.summary(dev)['probes'] == again and plain.numel() == 48 << 20
    assert placement.pick(1 << 20, dev).numel() == 1 << 20 and placement.summary(dev)['probes'] == again       # too small to probe
    # argument checks of the entry point
    lib = nat.lib()
    assert lib.rsa_placement_probe(None, 1 << 20, None, 1 << 20, 0, None) == -1 and b'null pointer' in lib.rsa_last_error()
    assert lib.rsa_placement_probe(buf.data_ptr(), 4096, buf.data_ptr(), 1 << 20, 0, None) == -1
    # a launch whose outputs are placed == the same launch on plain allocations
    N, U, d, B, n = 300_001, 4001, 128, 32768, 64                     # 32768 x 64 outputs: a 42 MB arena
    g = torch.Generator(device=DEV).manual_seed(5)
    item = torch.randn(N, d, device=DEV, generator=g) * 0.1
    user = torch.randn(U, d, device=DEV, generator=g) * 0.1
    uid = torch.randint(1, U, (B,), device=DEV, generator=g)
    pos = torch.randint(1, N, (B,), device=DEV, generator=g)
    torch.manual_seed(9)
    placed = ra.ops.fused_forward(item, user, n, query_index=uid, pos_ids=pos, fused_bpr=True, sampler=nat.SAMPLER_UNIFORM)
    torch.manual_seed(9)
    with placement.disabled():
        plain = ra.ops.fused_forward(item, user, n, query_index=uid, pos_ids=pos, fused_bpr=True, sampler=nat.SAMPLER_UNIFORM)
    for k in ('neg_ids', 'neg_score', 'dneg', 'row_loss', 'dpos', 'loss'):
        assert torch.equal(placed[k], plain[k]), k
    placement.release(dev)
    assert dev.index not in placement._state
