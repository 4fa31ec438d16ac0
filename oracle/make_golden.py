"""Generate tests/golden/*.npz by importing and RUNNING the real reference.

Runs only in the build container (needs /root/reference); the fixtures it
writes are data (inputs + the reference's outputs) and are committed, this
script is the provenance.  Usage:  python oracle/make_golden.py

Import recipe: stub ``nni``, ``torch.utils.tensorboard``
and ``torchmetrics`` (absent in the image, unused on this path), import
``recstudio.model`` before ``recstudio.ann.sampler``, run from a scratch cwd.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
REF = '/root/reference'


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    _stub('nni', get_next_parameter=lambda: {}, report_intermediate_result=lambda *a, **k: None,
          report_final_result=lambda *a, **k: None)

    class SummaryWriter:
        def __init__(self, *a, **k):
            self.log_dir = k.get('log_dir', a[0] if a else '.')

        def add_text(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass
    import torch.utils
    tb = _stub('torch.utils.tensorboard', SummaryWriter=SummaryWriter)
    torch.utils.tensorboard = tb
    tm = _stub('torchmetrics')
    tm.functional = _stub('torchmetrics.functional')
    sys.path.insert(0, REF)
    os.chdir(tempfile.mkdtemp(prefix='refcwd_'))
    import recstudio.model                                   # noqa: F401  (must come first)
    from recstudio.ann import sampler
    from recstudio.model import scorer, loss_func
    from recstudio.model.basemodel import BaseRetriever
    import recstudio.eval as ref_eval
    return sampler, scorer, loss_func, BaseRetriever, ref_eval


def np_(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def gen_score(scorer):
    out = {}
    g = torch.Generator().manual_seed(11)
    ip, cos, euc = scorer.InnerProductScorer(), scorer.CosineScorer(), scorer.EuclideanScorer()
    for d in (64, 128):
        B, n, L, N = 6, 5, 4, 9
        cases = {
            'bd_bd': (torch.randn(B, d, generator=g), torch.randn(B, d, generator=g)),
            'bd_bnd': (torch.randn(B, d, generator=g), torch.randn(B, n, d, generator=g)),
            'bld_bld': (torch.randn(B, L, d, generator=g), torch.randn(B, L, d, generator=g)),
            'bld_blnd': (torch.randn(B, L, d, generator=g), torch.randn(B, L, n, d, generator=g)),
            'bd_Nd': (torch.randn(B, d, generator=g), torch.randn(N, d, generator=g)),
        }
        for name, (q, it) in cases.items():
            k = f'd{d}_{name}'
            out[k + '_q'] = np_(q)
            out[k + '_items'] = np_(it)
            out[k + '_ip'] = np_(ip(q.clone(), it.clone()))
            out[k + '_cos'] = np_(cos(q.clone(), it.clone()))
            out[k + '_euc'] = np_(euc(q.clone(), it.clone()))
            out[k + '_norm2'] = np_(scorer.NormScorer(p=2)(q.clone(), it.clone()))              # scorer.py:56-66
            torch.manual_seed(100 + d)
            gmf = scorer.GMFScorer(d, bias=True, activation='relu')                              # scorer.py:69-86
            try:                      # the reference's GMF handles 2-D queries only (its unsqueeze(1) is wrong for [B, L, D])
                out[k + '_gmf'] = np_(gmf(q.clone(), it.clone()))
                out[k + '_gmf_w'] = np_(gmf.W.weight)
                out[k + '_gmf_b'] = np_(gmf.W.bias)
            except RuntimeError:
                pass
    np.savez_compressed(os.path.join(OUT, 'score.npz'), **out)


def gen_loss(loss_func):
    out = {}
    g = torch.Generator().manual_seed(12)

    def run(name, fn, **kw):
        leaves = {}
        args = {}
        for k, v in kw.items():
            if isinstance(v, torch.Tensor) and v.is_floating_point() and k in ('pos_score', 'neg_score', 'all_score'):
                v = v.clone().requires_grad_(True)
                leaves[k] = v
            args[k] = v
        val = fn(label=None, **args)
        out[name + '_loss'] = np_(val)
        if torch.isfinite(val):
            val.backward()
            for k, v in leaves.items():
                out[name + '_grad_' + k] = np_(v.grad)
        for k, v in kw.items():
            out[name + '_' + k] = np_(v)

    B, n, L = 7, 6, 3
    pos, neg = torch.randn(B, generator=g), torch.randn(B, n, generator=g)
    lpp = torch.log(torch.rand(B, generator=g))
    lnp = torch.log(torch.rand(B, n, generator=g))
    run('bpr_1d', loss_func.BPRLoss(), pos_score=pos, log_pos_prob=lpp, neg_score=neg, log_neg_prob=lnp)
    run('ssm_1d_f32', loss_func.SampledSoftmaxLoss(), pos_score=pos, log_pos_prob=lpp, neg_score=neg, log_neg_prob=lnp)
    run('ssm_1d_i64', loss_func.SampledSoftmaxLoss(), pos_score=pos, log_pos_prob=torch.zeros(B, dtype=torch.int64),
        neg_score=neg, log_neg_prob=torch.zeros(B, n, dtype=torch.int64))
    pos2, neg2 = torch.randn(B, L, generator=g), torch.randn(B, L, n, generator=g)
    lpp2 = torch.log(torch.rand(B, L, generator=g))
    lnp2 = torch.log(torch.rand(B, L, n, generator=g))
    run('bpr_2d', loss_func.BPRLoss(), pos_score=pos2, log_pos_prob=lpp2, neg_score=neg2, log_neg_prob=lnp2)
    run('ssm_2d', loss_func.SampledSoftmaxLoss(), pos_score=pos2, log_pos_prob=lpp2, neg_score=neg2, log_neg_prob=lnp2)
    # multi-positive rows sharing one negative set, one padded (-inf) positive  (loss_func.py:84-89)
    pos3 = torch.randn(B, L, generator=g)
    pos3[1, 2] = -float('inf')
    pos3[4, 0] = -float('inf')
    run('ssm_shared_pad', loss_func.SampledSoftmaxLoss(), pos_score=pos3, log_pos_prob=torch.zeros(B, L),
        neg_score=neg, log_neg_prob=lnp)
    # large magnitudes (stability of logsigmoid / logsumexp)
    run('bpr_big', loss_func.BPRLoss(), pos_score=pos * 30, log_pos_prob=lpp, neg_score=neg * 30, log_neg_prob=lnp)
    run('ssm_big', loss_func.SampledSoftmaxLoss(), pos_score=pos * 30, log_pos_prob=lpp, neg_score=neg * 30, log_neg_prob=lnp)
    # BinaryCrossEntropyLoss (SASRec's default, seq/sasrec.py:117-119), incl. padded (-inf) positives
    run('bce_1d', loss_func.BinaryCrossEntropyLoss(), pos_score=pos, log_pos_prob=lpp, neg_score=neg, log_neg_prob=lnp)
    pos4 = pos2.clone()
    pos4[0, 1] = -float('inf')
    pos4[5, 2] = -float('inf')
    run('bce_2d_pad', loss_func.BinaryCrossEntropyLoss(), pos_score=pos4, log_pos_prob=lpp2, neg_score=neg2, log_neg_prob=lnp2)
    alls = torch.randn(B, 50, generator=g)
    run('softmax_full', loss_func.SoftmaxLoss(), pos_score=pos, all_score=alls)
    # the other PairwiseLoss classes of loss_func.py (:93-97, :135-193).  Top1Loss (:66-78) is left out: its forward
    # modifies a sigmoid output in place, so the reference itself cannot back-propagate through it.
    for tag, (p_, lp_, n_, ln_) in (('1d', (pos, lpp, neg, lnp)), ('2d', (pos2, lpp2, neg2, lnp2))):
        run(f'wbpr_{tag}', loss_func.WeightedBPRLoss(), pos_score=p_, log_pos_prob=lp_, neg_score=n_, log_neg_prob=ln_)
        run(f'wbce_{tag}', loss_func.WeightedBinaryCrossEntropyLoss(), pos_score=p_, log_pos_prob=lp_, neg_score=n_,
            log_neg_prob=ln_)
        run(f'hinge_{tag}', loss_func.HingeLoss(margin=2), pos_score=p_, log_pos_prob=lp_, neg_score=n_, log_neg_prob=ln_)
        run(f'infonce_{tag}', loss_func.InfoNCELoss(), pos_score=p_, log_pos_prob=lp_, neg_score=n_, log_neg_prob=ln_)
        run(f'ccl_{tag}', loss_func.CCLLoss(margin=0.6, neg_weight=0.3), pos_score=p_, log_pos_prob=lp_, neg_score=n_,
            log_neg_prob=ln_)
    run('nce_1d', loss_func.NCELoss(), pos_score=pos, log_pos_prob=lpp, neg_score=neg, log_neg_prob=lnp)
    run('wbce_2d_pad', loss_func.WeightedBinaryCrossEntropyLoss(), pos_score=pos4, log_pos_prob=lpp2, neg_score=neg2,
        log_neg_prob=lnp2)
    run('wbpr_big', loss_func.WeightedBPRLoss(), pos_score=pos * 30, log_pos_prob=lpp, neg_score=neg * 30, log_neg_prob=lnp)
    run('hinge_inactive', loss_func.HingeLoss(margin=0.5), pos_score=pos + 5, log_pos_prob=lpp, neg_score=neg, log_neg_prob=lnp)
    # SoftmaxLoss second branch (loss_func.py:43-47): several positives per row [B, L] against one [B, N] score row,
    # padded (-inf) positives dropped from the row mean
    run('softmax_multi_pad', loss_func.SoftmaxLoss(), pos_score=pos3, all_score=alls)
    np.savez_compressed(os.path.join(OUT, 'loss.npz'), **out)


def gen_uniform(sampler):
    out = {}
    cases = [(1575, 8, 1, 2022), (1575, 512, 1, 7), (1000001, 16, 64, 3), (50, 4, 256, 5)]
    for i, (N, B, n, s) in enumerate(cases):
        us = sampler.UniformSampler(N)
        torch.manual_seed(s)
        q = torch.zeros(B, 4)
        pos = torch.arange(B) % (N - 1) + 1
        pp, neg, npb = us(q, n, pos)
        out[f'c{i}_meta'] = np.array([N, B, n, s])
        out[f'c{i}_neg'] = np_(neg)
        assert pp.dtype == torch.int64 and npb.dtype == torch.int64 and neg.dtype == torch.int64
        assert int(pp.abs().sum()) == 0 and int(npb.abs().sum()) == 0
    # 3-D query and int query, no pos_items
    us = sampler.UniformSampler(300)
    torch.manual_seed(9)
    neg, npb = us(torch.zeros(3, 5, 4), 7)
    out['q3d_neg'] = np_(neg)
    torch.manual_seed(9)
    neg, npb = us(15, 7, device=torch.device('cpu'))
    out['qint_neg'] = np_(neg)
    # MaskedUniformSampler / uniform_sample_masked_hist (sampler.py:117-147, :187-214)
    g = torch.Generator().manual_seed(31)
    Nm, Bm, Lh, nm = 60, 9, 12, 40
    hist = torch.zeros(Bm, Lh, dtype=torch.int64)
    for b in range(Bm):
        m = int(torch.randint(0, Lh + 1, (1,), generator=g))
        hist[b, :m] = torch.randperm(Nm - 1, generator=g)[:m] + 1
    hist[3] = hist[3][torch.randperm(Lh, generator=g)]        # padding not at the end
    torch.manual_seed(77)
    u = torch.rand(Bm, nm)
    torch.manual_seed(77)
    neg = sampler.uniform_sample_masked_hist(num_items=Nm - 1, num_neg=nm, user_hist=hist)
    torch.manual_seed(78)
    u3 = torch.rand(Bm, 3 * 5)
    torch.manual_seed(78)
    neg3 = sampler.uniform_sample_masked_hist(num_items=Nm - 1, num_neg=5, user_hist=hist, num_query_per_user=3)
    out.update(mask_N=Nm, mask_hist=np_(hist), mask_u=np_(u), mask_neg=np_(neg), mask_u3=np_(u3), mask_neg3=np_(neg3))
    ms = sampler.MaskedUniformSampler(Nm)
    torch.manual_seed(79)
    pp, ng, npb = ms(torch.zeros(Bm, 4), nm, torch.ones(Bm, dtype=torch.int64), hist)
    out.update(mask_fw_neg=np_(ng), mask_fw_negprob=np_(npb), mask_fw_posprob=np_(pp))
    np.savez_compressed(os.path.join(OUT, 'uniform.npz'), **out)


def gen_popular(sampler):
    out = {}
    g = torch.Generator().manual_seed(13)
    N = 2000
    counts = (torch.rand(N, generator=g) ** 4 * 500).long()
    counts[5:9] = 0
    counts[100:104] = 17                                     # ties
    for mode in (0, 1, 2):
        ps = sampler.PopularSamplerModel(counts.clone(), mode=mode)
        out[f'm{mode}_pop_prob'] = np_(ps.pop_prob)
        out[f'm{mode}_table'] = np_(ps.table)
        tbl = ps.table
        u = torch.rand(4096, generator=g)
        edge = torch.tensor([0.0, float(tbl[0]), float(tbl[1]), float(tbl[57]), float(tbl[-2]),
                             float(np.nextafter(np.float32(tbl[57]), np.float32(1))),
                             float(np.nextafter(np.float32(tbl[57]), np.float32(0))),
                             1.0 - 2.0 ** -24, 2.0 ** -33], dtype=torch.float32)
        u = torch.cat([edge, u])
        ids = torch.searchsorted(ps.table, u)
        ids_c = ids.clamp(max=N - 1)
        out[f'm{mode}_u'] = np_(u)
        out[f'm{mode}_ids'] = np_(ids)
        out[f'm{mode}_logp'] = np_(ps.compute_item_p(None, ids_c))
        torch.manual_seed(21 + mode)
        pos = torch.randint(0, N, (9,))
        pp, neg, npb = ps(torch.zeros(9, 4), 33, pos)
        out[f'm{mode}_fw_pos'] = np_(pos)
        out[f'm{mode}_fw_pp'] = np_(pp)
        out[f'm{mode}_fw_neg'] = np_(neg)
        out[f'm{mode}_fw_np'] = np_(npb)
    out['counts'] = np_(counts)
    # a larger table: cumsum rounding behaviour (float64 accumulation on CPU)
    N2 = 120000
    counts2 = (torch.rand(N2, generator=g) ** 6 * 3000).long()
    ps = sampler.PopularSamplerModel(counts2.clone(), mode=0)
    out['big_counts'] = np_(counts2)
    out['big_table_tail'] = np_(ps.table[-64:])
    out['big_table_probe_idx'] = np.arange(0, N2, 997)
    out['big_table_probe'] = np_(ps.table[::997])
    u = torch.rand(2048, generator=g)
    out['big_u'] = np_(u)
    out['big_ids'] = np_(torch.searchsorted(ps.table, u))
    np.savez_compressed(os.path.join(OUT, 'popular.npz'), **out)


class _FixedSampler:
    pass


def gen_forward(sampler, scorer, loss_func, BaseRetriever):
    """Tiny BaseRetriever driven through training_step with a sampler that returns given ids."""
    out = {}
    U, N, d, B, n = 50, 200, 16, 8, 4

    def make_fixed(ids, lpp, lnp):
        class Fixed(sampler.Sampler):
            def forward(self, query, num_neg, pos_items=None):
                return lpp, ids, lnp
        return Fixed(N)

    g = torch.Generator().manual_seed(14)
    for tag, sc, lossf in (('bpr_ip', scorer.InnerProductScorer(), loss_func.BPRLoss()),
                           ('ssm_ip', scorer.InnerProductScorer(), loss_func.SampledSoftmaxLoss()),
                           ('bpr_cos', scorer.CosineScorer(), loss_func.BPRLoss()),
                           ('softmax_ip', scorer.InnerProductScorer(), loss_func.SoftmaxLoss())):
        item = torch.nn.Embedding(N, d, padding_idx=0)
        user = torch.nn.Embedding(U, d, padding_idx=0)
        with torch.no_grad():
            item.weight.copy_(torch.randn(N, d, generator=g) * 0.3)
            item.weight[0] = 0
            user.weight.copy_(torch.randn(U, d, generator=g) * 0.3)
            user.weight[0] = 0
        uid = torch.randint(1, U, (B,), generator=g)
        pos = torch.randint(1, N, (B,), generator=g)
        neg = torch.randint(1, N, (B, n), generator=g)
        neg[0, 0] = neg[0, 1]                                # duplicate negative in a row
        neg[1, 2] = pos[1]                                   # negative == positive
        lpp = torch.log(torch.rand(B, generator=g))
        lnp = torch.log(torch.rand(B, n, generator=g))
        full = tag.startswith('softmax')
        kw = dict(item_encoder=item, query_encoder=user, scorer=sc, loss=lossf)
        if not full:
            kw['sampler'] = make_fixed(neg, lpp, lnp)
        model = BaseRetriever(None, **kw)
        model.fiid, model.fuid, model.frating = 'item_id', 'user_id', 'rating'
        model.item_fields = {'item_id'}
        model.query_fields = {'user_id'}
        model.neg_count = n
        batch = {'user_id': uid, 'item_id': pos, 'rating': torch.ones(B)}
        fw = model.forward(batch, full_score=full)
        loss = model.training_step(batch)
        loss.backward()
        out[tag + '_item_w'] = np_(item.weight)
        out[tag + '_user_w'] = np_(user.weight)
        out[tag + '_uid'], out[tag + '_pos'], out[tag + '_neg'] = np_(uid), np_(pos), np_(neg)
        out[tag + '_lpp'], out[tag + '_lnp'] = np_(lpp), np_(lnp)
        out[tag + '_pos_score'] = np_(fw['score']['pos_score'])
        if full:
            out[tag + '_all_score'] = np_(fw['score']['all_score'])
        else:
            out[tag + '_neg_score'] = np_(fw['score']['neg_score'])
        out[tag + '_loss'] = np_(loss)
        out[tag + '_item_grad'] = np_(item.weight.grad)
        out[tag + '_user_grad'] = np_(user.weight.grad)
    np.savez_compressed(os.path.join(OUT, 'forward.npz'), **out)


def gen_topk(scorer, BaseRetriever, ref_eval):
    out = {}
    U, N, d, B, k = 30, 120, 16, 6, 10
    g = torch.Generator().manual_seed(15)
    item = torch.nn.Embedding(N, d, padding_idx=0)
    user = torch.nn.Embedding(U, d, padding_idx=0)
    with torch.no_grad():
        item.weight.copy_(torch.randn(N, d, generator=g))
        item.weight[0] = 0
        user.weight.copy_(torch.randn(U, d, generator=g))
    model = BaseRetriever(None, item_encoder=item, query_encoder=user, scorer=scorer.InnerProductScorer())
    model.fiid, model.fuid, model.frating = 'item_id', 'user_id', 'rating'
    model.item_fields = {'item_id'}
    model.query_fields = {'user_id'}
    model.config['eval']['topk'] = k
    model._update_item_vector()
    uid = torch.randint(1, U, (B,), generator=g)
    hist = torch.zeros(B, 7, dtype=torch.int64)
    for b in range(B):
        m = int(torch.randint(1, 8, (1,), generator=g))
        hist[b, :m] = torch.randperm(N - 1, generator=g)[:m] + 1
    # make sure some history items are top scorers so that masking matters
    with torch.no_grad():
        top = (user.weight[uid] @ item.weight[1:].T).topk(3).indices + 1
    hist[:, 0] = top[:, 0]
    hist[2, 1] = top[2, 1]
    batch = {'user_id': uid, 'user_hist': hist}
    with torch.no_grad():
        score, items = model.topk(batch, k, hist)
        score_nh, items_nh = model.topk(batch, k, None)
    out.update(item_w=np_(item.weight), user_w=np_(user.weight), uid=np_(uid), hist=np_(hist),
               score=np_(score), items=np_(items), score_nohist=np_(score_nh), items_nohist=np_(items_nh))
    # _test_step: 1-D target and 2-D target
    tgt1 = items[:, 3].clone()
    tgt1[0] = 1 if 1 not in items[0] else 2
    batch1 = {'user_id': uid, 'user_hist': hist, 'item_id': tgt1, 'rating': torch.ones(B)}
    tgt2 = torch.zeros(B, 3, dtype=torch.int64)
    tgt2[:, 0] = items[:, 1]
    tgt2[:3, 1] = items[:3, 7]
    rat2 = (tgt2 > 0).float() * 4
    batch2 = {'user_id': uid, 'user_hist': hist, 'item_id': tgt2, 'rating': rat2}
    metrics = ['ndcg', 'recall', 'precision', 'map', 'mrr', 'hit']
    with torch.no_grad():
        m1, bs1 = model._test_step(batch1, metrics, [5, 10])
        m2, bs2 = model._test_step(batch2, metrics, [5, 10])
    out['tgt1'], out['tgt2'], out['rat2'] = np_(tgt1), np_(tgt2), np_(rat2)
    for key, v in m1.items():
        out['m1_' + key] = np_(v)
    for key, v in m2.items():
        out['m2_' + key] = np_(v)
    np.savez_compressed(os.path.join(OUT, 'topk.npz'), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    sampler, scorer, loss_func, BaseRetriever, ref_eval = import_reference()
    torch.set_num_threads(1)
    gen_score(scorer)
    gen_loss(loss_func)
    gen_uniform(sampler)
    gen_popular(sampler)
    gen_forward(sampler, scorer, loss_func, BaseRetriever)
    gen_topk(scorer, BaseRetriever, ref_eval)
    if os.path.exists(os.path.join(HERE, 'make_golden_data.py')):
        from make_golden_data import gen_data
        gen_data(OUT)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    sys.path.insert(0, HERE)
    main()
