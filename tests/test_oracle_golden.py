"""CPU: the oracle (oracle/) against fixtures recorded from the real reference
(oracle/make_golden.py) and against published known-answer vectors."""
import numpy as np
import torch

import oracle
from oracle import philox

T = torch.from_numpy


def test_philox_kat():
    # Random123 kat_vectors, philox4x32 10 rounds
    def run(c, k):
        return [int(x) for x in philox.philox4x32_10(np.array(c, dtype=np.uint32), np.array(k, dtype=np.uint32))]
    assert run([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert run([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert run([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_device_stream_shape_and_range():
    G = philox.rng_grid_threads(1000, 256, 2048)
    assert G == 1024
    assert philox.rng_grid_threads(10 ** 7, 256, 2048) == 2048 * 256
    assert philox.rng_counter_offset(1000, G, 4) == 4
    assert philox.rng_counter_offset(5 * 2048 * 256, 2048 * 256, 4) == 8
    ids = philox.device_randint(2022, 0, 5000, 1, 1575, philox.rng_grid_threads(5000, 256, 2048))
    assert ids.min() >= 1 and ids.max() <= 1574 and ids.dtype == np.int64
    # 64-bit path (range >= 2**28)
    ids = philox.device_randint(7, 8, 3000, 1, 2 ** 30, 1024)
    assert ids.min() >= 1 and ids.max() < 2 ** 30
    u = philox.device_rand(2022, 4, 5000, 5120)
    assert u.dtype == np.float32 and u.min() >= 0.0 and u.max() < 1.0
    # different offsets / elements decorrelate; same state reproduces
    assert np.array_equal(u, philox.device_rand(2022, 4, 5000, 5120))
    assert not np.array_equal(u, philox.device_rand(2022, 8, 5000, 5120))


def test_scorers(golden):
    g = golden('score')
    for d in (64, 128):
        for case in ('bd_bd', 'bd_bnd', 'bld_bld', 'bld_blnd', 'bd_Nd'):
            k = f'd{d}_{case}'
            q, it = T(g[k + '_q']), T(g[k + '_items'])
            np.testing.assert_allclose(oracle.inner_product_score(q, it).numpy(), g[k + '_ip'], rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(oracle.cosine_score(q, it).numpy(), g[k + '_cos'], rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(oracle.euclidean_score(q, it).numpy(), g[k + '_euc'], rtol=1e-5, atol=1e-4)
            np.testing.assert_allclose(oracle.norm_score(q, it).numpy(), g[k + '_norm2'], rtol=1e-6, atol=1e-6)
            if k + '_gmf' in g:
                got = oracle.gmf_score(q, it, T(g[k + '_gmf_w']), T(g[k + '_gmf_b']))
                np.testing.assert_allclose(got.numpy(), g[k + '_gmf'], rtol=1e-5, atol=1e-6)


def test_losses(golden):
    g = golden('loss')

    def grads(fn, *names_and_vals):
        leaves = [v.clone().requires_grad_(True) if req else v for v, req in names_and_vals]
        val = fn(*leaves)
        if torch.isfinite(val):
            val.backward()
        return val, [x.grad for x, (_, req) in zip(leaves, names_and_vals) if req]

    for name in ('bpr_1d', 'bpr_2d', 'bpr_big'):
        val, (gp, gn) = grads(oracle.bpr_loss, (T(g[name + '_pos_score']), True), (T(g[name + '_neg_score']), True))
        np.testing.assert_allclose(val.detach().numpy(), g[name + '_loss'], rtol=1e-6)
        np.testing.assert_allclose(gp.numpy(), g[name + '_grad_pos_score'], rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(gn.numpy(), g[name + '_grad_neg_score'], rtol=1e-5, atol=1e-8)
    for name in ('ssm_1d_f32', 'ssm_1d_i64', 'ssm_2d', 'ssm_shared_pad', 'ssm_big'):
        fn = lambda p, n: oracle.sampled_softmax_loss(p, T(g[name + '_log_pos_prob']), n, T(g[name + '_log_neg_prob']))
        val, (gp, gn) = grads(fn, (T(g[name + '_pos_score']), True), (T(g[name + '_neg_score']), True))
        np.testing.assert_allclose(val.detach().numpy(), g[name + '_loss'], rtol=1e-6, equal_nan=True)
        if name + '_grad_pos_score' in g:
            np.testing.assert_allclose(gp.numpy(), g[name + '_grad_pos_score'], rtol=1e-5, atol=1e-8)
            np.testing.assert_allclose(gn.numpy(), g[name + '_grad_neg_score'], rtol=1e-5, atol=1e-8)
    for name in ('bce_1d', 'bce_2d_pad'):
        val, (gp, gn) = grads(oracle.bce_loss, (T(g[name + '_pos_score']), True), (T(g[name + '_neg_score']), True))
        np.testing.assert_allclose(val.detach().numpy(), g[name + '_loss'], rtol=1e-6)
        np.testing.assert_allclose(gp.numpy(), g[name + '_grad_pos_score'], rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(gn.numpy(), g[name + '_grad_neg_score'], rtol=1e-5, atol=1e-8)
    # the other PairwiseLoss classes (loss_func.py:93-97, :135-193)
    others = {
        'wbpr': lambda name: (lambda p, n: oracle.weighted_bpr_loss(p, n, T(g[name + '_log_neg_prob']))),
        'wbce': lambda name: (lambda p, n: oracle.weighted_bce_loss(p, n, T(g[name + '_log_neg_prob']))),
        'hinge': lambda name: (lambda p, n: oracle.hinge_loss(p, n, 0.5 if name == 'hinge_inactive' else 2.0)),
        'infonce': lambda name: oracle.info_nce_loss,
        'ccl': lambda name: (lambda p, n: oracle.ccl_loss(p, n, 0.6, 0.3)),
        'nce': lambda name: (lambda p, n: oracle.nce_loss(p, T(g[name + '_log_pos_prob']), n, T(g[name + '_log_neg_prob']))),
    }
    for name in ('wbpr_1d', 'wbpr_2d', 'wbpr_big', 'wbce_1d', 'wbce_2d', 'wbce_2d_pad', 'hinge_1d', 'hinge_2d',
                 'hinge_inactive', 'infonce_1d', 'infonce_2d', 'ccl_1d', 'ccl_2d', 'nce_1d'):
        fn = others[name.split('_')[0]](name)
        val, (gp, gn) = grads(fn, (T(g[name + '_pos_score']), True), (T(g[name + '_neg_score']), True))
        np.testing.assert_allclose(val.detach().numpy(), g[name + '_loss'], rtol=1e-6, err_msg=name)
        np.testing.assert_allclose(gp.numpy(), g[name + '_grad_pos_score'], rtol=1e-5, atol=1e-8, err_msg=name)
        np.testing.assert_allclose(gn.numpy(), g[name + '_grad_neg_score'], rtol=1e-5, atol=1e-8, err_msg=name)
    val, (gp, ga) = grads(oracle.softmax_loss, (T(g['softmax_full_pos_score']), True), (T(g['softmax_full_all_score']), True))
    np.testing.assert_allclose(val.detach().numpy(), g['softmax_full_loss'], rtol=1e-6)
    np.testing.assert_allclose(ga.numpy(), g['softmax_full_grad_all_score'], rtol=1e-5, atol=1e-8)
    # second branch (several positives per row, -inf padding dropped)
    val, (gp, ga) = grads(oracle.softmax_loss, (T(g['softmax_multi_pad_pos_score']), True), (T(g['softmax_multi_pad_all_score']), True))
    np.testing.assert_allclose(val.detach().numpy(), g['softmax_multi_pad_loss'], rtol=1e-6)
    np.testing.assert_allclose(ga.numpy(), g['softmax_multi_pad_grad_all_score'], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(gp.numpy(), g['softmax_multi_pad_grad_pos_score'], rtol=1e-5, atol=1e-8)


def test_uniform_sampler_cpu_stream(golden):
    g = golden('uniform')
    for i in range(4):
        N, B, n, s = (int(x) for x in g[f'c{i}_meta'])
        torch.manual_seed(s)
        us = oracle.UniformSampler(N)
        pp, neg, npb = us.forward(torch.zeros(B, 4), n, torch.ones(B, dtype=torch.int64))
        assert np.array_equal(neg.numpy(), g[f'c{i}_neg'])
        assert neg.min() >= 1 and neg.max() <= N - 1
        assert pp.dtype == torch.int64 and npb.dtype == torch.int64 and not npb.any()
    us = oracle.UniformSampler(300)
    torch.manual_seed(9)
    neg, _ = us.forward(torch.zeros(3, 5, 4), 7)
    assert np.array_equal(neg.numpy(), g['q3d_neg']) and neg.shape == (3, 5, 7)
    torch.manual_seed(9)
    neg, _ = us.forward(15, 7)
    assert np.array_equal(neg.numpy(), g['qint_neg'])


def test_masked_uniform_sampler(golden):
    g = golden('uniform')
    N, hist = int(g['mask_N']), T(g['mask_hist'])
    neg = oracle.masked_uniform_from_u(N - 1, g['mask_neg'].shape[1], hist, g['mask_u'])
    assert np.array_equal(neg.numpy(), g['mask_neg'])
    neg3 = oracle.masked_uniform_from_u(N - 1, 5, hist, g['mask_u3'], num_query_per_user=3)
    assert np.array_equal(neg3.numpy(), g['mask_neg3'])
    for b in range(hist.shape[0]):          # never a history item, never padding, always in range
        assert not set(neg[b].tolist()) & set(hist[b].tolist())
        assert neg[b].min() >= 1 and neg[b].max() <= N - 1
    torch.manual_seed(79)
    u = torch.rand(hist.shape[0], g['mask_fw_neg'].shape[1])
    assert np.array_equal(oracle.masked_uniform_from_u(N - 1, u.shape[1], hist, u).numpy(), g['mask_fw_neg'])
    assert not g['mask_fw_negprob'].any() and not g['mask_fw_posprob'].any()    # -log(1) = 0, int64


def test_popular_sampler(golden):
    g = golden('popular')
    counts = T(g['counts'])
    for mode in (0, 1, 2):
        ps = oracle.PopularSamplerModel(counts.clone(), mode=mode)
        assert np.array_equal(ps.pop_prob.numpy(), g[f'm{mode}_pop_prob'])     # exact fp32 bits
        assert np.array_equal(ps.table.numpy(), g[f'm{mode}_table'])
        ids = ps.ids_from_uniform(g[f'm{mode}_u'], clamp=False)
        assert np.array_equal(ids.numpy(), g[f'm{mode}_ids'])
        logp = ps.compute_item_p(ids.clamp(max=len(counts) - 1))
        np.testing.assert_array_equal(logp.numpy(), g[f'm{mode}_logp'])
        torch.manual_seed(21 + mode)
        pos = torch.randint(0, len(counts), (9,))
        assert np.array_equal(pos.numpy(), g[f'm{mode}_fw_pos'])
        pp, neg, npb = ps.forward(torch.zeros(9, 4), 33, pos)
        assert np.array_equal(neg.numpy(), g[f'm{mode}_fw_neg'])
        np.testing.assert_array_equal(pp.numpy(), g[f'm{mode}_fw_pp'])
        np.testing.assert_array_equal(npb.numpy(), g[f'm{mode}_fw_np'])
    # torch's CPU fp32 .sum() (sampler.py:239) is thread-count dependent in the last ulp above the
    # parallel grain (32768 elements); the fixture was recorded at 1 thread.
    nt = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        ps = oracle.PopularSamplerModel(T(g['big_counts']), mode=0)
    finally:
        torch.set_num_threads(nt)
    assert np.array_equal(ps.table[-64:].numpy(), g['big_table_tail'])
    assert np.array_equal(ps.table.numpy()[g['big_table_probe_idx']], g['big_table_probe'])
    assert np.array_equal(ps.ids_from_uniform(g['big_u'], clamp=False).numpy(), g['big_ids'])


def test_forward_and_dense_grads(golden):
    g = golden('forward')
    for tag, loss, cos in (('bpr_ip', 'bpr', False), ('ssm_ip', 'ssm', False), ('bpr_cos', 'bpr', True),
                           ('softmax_ip', 'softmax', False)):
        iw, uw = T(g[tag + '_item_w']), T(g[tag + '_user_w'])
        uid, pos, neg = T(g[tag + '_uid']), T(g[tag + '_pos']), T(g[tag + '_neg'])
        val, ps, ns, gi, gu = oracle.dense_grads(iw, uw, uid, pos, neg, loss=loss, cosine=cos,
                                                 log_pos_prob=T(g[tag + '_lpp']), log_neg_prob=T(g[tag + '_lnp']))
        np.testing.assert_allclose(ps.numpy(), g[tag + '_pos_score'], rtol=1e-5, atol=1e-6)
        if loss != 'softmax':
            np.testing.assert_allclose(ns.numpy(), g[tag + '_neg_score'], rtol=1e-5, atol=1e-6)
            p2, n2 = oracle.retriever_forward(iw, uw[uid], pos, neg, cosine=cos)
            np.testing.assert_allclose(n2.numpy(), g[tag + '_neg_score'], rtol=1e-5, atol=1e-6)
        else:
            p2, a2 = oracle.retriever_forward(iw, uw[uid], pos, None, full=True)
            np.testing.assert_allclose(a2.numpy(), g[tag + '_all_score'], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(val.numpy(), g[tag + '_loss'], rtol=1e-6)
        np.testing.assert_allclose(gi.numpy(), g[tag + '_item_grad'], rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(gu.numpy(), g[tag + '_user_grad'], rtol=1e-5, atol=1e-8)
        assert not gi[0].any() and not g[tag + '_item_grad'][0].any()      # padding row: no gradient


def test_topk_and_metrics(golden):
    g = golden('topk')
    iw, uw, uid, hist = T(g['item_w']), T(g['user_w']), T(g['uid']), T(g['hist'])
    k = g['items'].shape[1]
    score, items = oracle.topk_with_history(uw[uid], iw, k, hist)
    assert np.array_equal(items.numpy(), g['items'])
    np.testing.assert_allclose(score.numpy(), g['score'], rtol=1e-6)
    score, items_nh = oracle.topk_with_history(uw[uid], iw, k, None)
    assert np.array_equal(items_nh.numpy(), g['items_nohist'])
    for tgt, rating, pre in ((T(g['tgt1']), torch.ones(len(uid), 1), 'm1_'), (T(g['tgt2']), T(g['rat2']), 'm2_')):
        hits = oracle.test_step_hits(tgt, items)
        for cutoff in (5, 10):
            m = oracle.rank_metrics(hits, rating, cutoff)
            for name, v in m.items():
                np.testing.assert_allclose(v.numpy(), g[f'{pre}{name}@{cutoff}'], rtol=1e-6, err_msg=f'{pre}{name}@{cutoff}')


def test_seq_gather_restatement():
    iw = torch.randn(20, 8)
    iw[0] = 0
    flat = torch.tensor([3, 4, 5, 6, 7, 8, 9, 1, 2])
    ids, rows, lens = oracle.seq_gather(iw, flat, [0, 2, 7], [2, 7, 8], 6)
    assert ids.tolist() == [[3, 4, 0, 0, 0, 0], [5, 6, 7, 8, 9, 0], [1, 0, 0, 0, 0, 0]]
    assert torch.equal(rows[1, 4], iw[9]) and not rows[0, 2:].any() and lens.tolist() == [2, 5, 1]


def test_sasrec_tower(golden):
    """oracle.sasrec_query (sasrec.py:37-67 restated with stock torch modules) against the recorded run of the reference's
    own SASRecQueryEncoder (tests/golden/sasrec.npz): tower output, scores and both losses."""
    g = golden('sasrec')
    state = {k[3:]: T(g[k]) for k in g.files if k.startswith('w::')}
    hist, seqlen = T(g['hist']), T(g['seqlen'])
    with torch.no_grad():
        q = oracle.sasrec_query(state, hist, seqlen, int(g['heads']), int(g['hidden']), int(g['layers']))
    np.testing.assert_allclose(q.numpy(), g['ssm_query'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(q.numpy(), g['bce_query'], rtol=1e-5, atol=1e-6)
    item_w = state['item_encoder.weight']
    pos, neg = T(g['pos']), T(g['neg'])
    ps, ns = oracle.retriever_forward(item_w, q, pos, neg)
    np.testing.assert_allclose(ns.numpy(), g['ssm_neg_score'], rtol=1e-5, atol=1e-6)
    ssm = oracle.sampled_softmax_loss(ps, T(g['log_pos']), ns, T(g['log_neg']))
    np.testing.assert_allclose(ssm.item(), float(g['ssm_loss']), rtol=1e-6)
    np.testing.assert_allclose(oracle.bce_loss(ps, ns).item(), float(g['bce_loss']), rtol=1e-6)
