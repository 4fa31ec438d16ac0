#!/usr/bin/env python
"""Throughput of the thing users run: ``BaseRetriever.fit`` (VERDICT r4 "next" #1).

Two figures, both through the unchanged ``fit`` (loader -> training step -> optimizer; the reference loop is
recstudio/model/basemodel/recommender.py:560-650):

  c1   BASELINE.json configs[0]: stock ``BPR`` on the committed ml-100k fixture (d = 64, B = 512, n = 1, dense Adam, validation
       every epoch with eval batch 20) -- steady-state train s/epoch and valid s/epoch, next to the reference's published
       log (/root/reference/README.md:197-204: 0.418 s train, 0.18-0.32 s valid on the authors' GPU).
  c2   BASELINE.json configs[1] shape: a synthetic interaction stream (users uniform, items Zipf(1) over a permuted id space)
       as a ``TripletDataset``, device-resident loader, popularity sampler, n = 64, d = 128, ``train.fused_optimizer: 'sgd'``,
       B in {4096, 65 536}: M triplets/s THROUGH THE LOOP, next to the same step issued back to back by a bare loop over the
       stepper (what bench.py's ``train_step.sgd_step_prefetched_ms`` times).

  c3   BASELINE.json configs[2]: ``SASRec`` (d = 128, max_seq_len = 50, 2 layers, 2 heads) with ``SampledSoftmaxLoss``, popularity
       sampler n = 256, on a synthetic ``SeqDataset``-shaped stream (users with a taste cluster; device loader -> ``rsa_seg_gather``
       -> stock Transformer -> the fused sample + score + SampledSoftmax launch -> autograd -> torch Adam): ms/step THROUGH
       ``fit``, and next to it where a step goes -- {history gather, Transformer fwd + bwd (stock torch), sample + score + loss,
       backward scatters, optimizer} timed piece by piece on one batch -- i.e. the share of a SASRec step the hot path holds.

Every figure carries ``train_loss_first_last`` and the loop ASSERTS that the loss moved (c2, c3: last < first - 0.01): a loop that
never moves the loss is no evidence that the loop trains.

Stand-alone: ``python tools/bench_fit.py [--items N] [--inter M]`` prints one JSON object.  bench.py imports ``fit_figures``.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _ml100k(ra):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'data_ml100k.npz'))
    return ra.TripletDataset('ml-100k', {'low_rating_thres': 3.0},
                             _interactions=(g['raw_user'].astype(str), g['raw_item'].astype(str),
                                            g['raw_rating'].astype(np.float64), g['raw_time'].astype(np.float64)))


def _steady(xs):
    """median of the epochs after the first two (allocator warm-up, first-launch module loads)"""
    xs = sorted(xs[2:] if len(xs) > 3 else xs)
    return xs[len(xs) // 2]


def fit_c1(ra, epochs=10):
    model = ra.BPR({'eval': {'batch_size': 20}, 'train': {'epochs': epochs, 'early_stop_patience': 10 ** 6}})
    ds = _ml100k(ra)
    trn, val, tst = ds.build(split_ratio=[0.8, 0.1, 0.1], shuffle=True, split_mode='user_entry')
    t0 = time.perf_counter()
    model.fit(trn, val)
    wall = time.perf_counter() - t0
    h = model.history
    steps = (len(trn) + 511) // 512
    tr, va = _steady([e['train_time'] for e in h]), _steady([e['valid_time'] for e in h])
    return {'train_s_per_epoch': round(tr, 4), 'valid_s_per_epoch': round(va, 4), 'epochs': len(h),
            'steps_per_epoch': steps, 'train_ms_per_step': round(tr / steps * 1e3, 3),
            'first_epoch_train_s': round(h[0]['train_time'], 3), 'fit_wall_s': round(wall, 2),
            'train_loss_last': round(h[-1]['train_loss'], 4),
            'reference_published_s_per_epoch': {'train': 0.418, 'valid': [0.18, 0.32]}}


def _zipf_ranks(n, count, dev, g):
    """`count` draws of a rank in [0, n) with probability ~ 1 / (rank + 1), on the device"""
    cdf = torch.cumsum(1.0 / torch.arange(1, n + 1, dtype=torch.float64, device=dev), 0)
    cdf /= cdf[-1].clone()
    u = torch.rand(count, dtype=torch.float64, device=dev, generator=g)
    return torch.searchsorted(cdf, u).clamp_(max=n - 1)


def synthetic_dataset(ra, dev, n_users, n_items, n_inter, seed=1, clusters=1024, in_cluster=1.0):
    """The configs[1] stream as a TripletDataset whose ids are already mapped (what ``load_cache`` yields): users uniform,
    items Zipf(1) over a permuted id space, drawn on the device; every interaction is a training sample.  So that there is
    something to LEARN (a stream whose positives and popularity-sampled negatives are identically distributed pins BPR at
    ln 2), a user belongs to one of ``clusters`` taste clusters and draws ``in_cluster`` of its items from that cluster's slice of
    the catalog (Zipf inside it), the rest from the whole catalog.  Default 1.0 -- the item marginal is then ``clusters``
    interleaved Zipf(1) slices: with a share of GLOBAL Zipf draws the head items collect tens of thousands of (popularity-
    sampled) negatives per B = 65 536 step, and plain SGD on them is only stable at a rate at which nothing else moves."""
    g = torch.Generator(device=dev).manual_seed(seed)
    perm = torch.randperm(n_items - 1, device=dev, generator=g)
    users = torch.randint(1, n_users, (n_inter,), device=dev, generator=g)
    per = (n_items - 1) // clusters
    own = torch.rand(n_inter, device=dev, generator=g) < in_cluster
    glob = _zipf_ranks(n_items - 1, n_inter, dev, g)
    loc = (users % clusters) + clusters * _zipf_ranks(per, n_inter, dev, g)          # ranks c, c + C, c + 2C, ...: the cluster's slice
    items = perm[torch.where(own, loc, glob)] + 1
    ds = ra.TripletDataset.from_mapped_ids(users.cpu(), items.cpu(), n_users=n_users, n_items=n_items, name='synthetic')
    del perm, own, glob, loc
    return ds


def synthetic_seq_dataset(ra, dev, n_users, n_items, max_seq_len=50, len_lo=5, len_hi=76, seed=2, clusters=1000, in_cluster=0.8):
    """A SeqDataset-shaped stream with mapped ids (configs[2]): every user has a time-ordered sequence of len_lo .. len_hi-1
    items; a user has a taste cluster (items c, c + C, c + 2C, ... of the catalog, Zipf inside it) and takes ``in_cluster`` of
    its items from it.  Every prefix of length >= 1 is a sample [user, start, end) -> target = item at ``end`` (dataset.py:1369-1445),
    windows cut to the last ``max_seq_len`` items."""
    g = torch.Generator(device=dev).manual_seed(seed)
    lens = torch.randint(len_lo, len_hi, (n_users,), device=dev, generator=g)
    total = int(lens.sum())
    uid = torch.repeat_interleave(torch.arange(1, n_users + 1, device=dev), lens)
    first = torch.repeat_interleave(torch.cumsum(lens, 0) - lens, lens)
    per = (n_items - 1) // clusters
    own = torch.rand(total, device=dev, generator=g) < in_cluster
    cl = (uid * 7919) % clusters
    loc = cl + clusters * _zipf_ranks(per, total, dev, g)
    glob = _zipf_ranks(n_items - 1, total, dev, g)
    items = torch.where(own, loc, glob) + 1
    ds = ra.SeqDataset.from_mapped_ids(uid.cpu(), items.cpu(), n_users=n_users + 1, n_items=n_items,
                                       config={'max_seq_len': max_seq_len}, name='synthetic-seq')
    posn = torch.arange(total, device=dev)
    keep = posn > first
    ds.data_index = torch.stack([uid, torch.maximum(first, posn - max_seq_len), posn], 1)[keep].cpu()
    return ds


def fit_c2(ra, dev, ds, B, epochs, lr_per_sample=0.25, d=128, n=64):
    """``lr_per_sample``: the kernels apply plain SGD to the MEAN loss over the batch, so the step a single interaction makes is
    lr / B: lr = lr_per_sample * B keeps it the same at every batch size (and large enough that the loss of a few epochs over
    a stream with 16 interactions per user moves visibly)."""
    sampler = ra.PopularSamplerModel(ds.item_freq)
    lr = lr_per_sample * B
    conf = {'model': {'embed_dim': d},
            'train': {'negative_count': n, 'batch_size': B, 'fused_optimizer': 'sgd', 'epochs': epochs, 'learning_rate': lr,
                      'init_method': 'normal', 'early_stop_patience': 10 ** 6}}
    model = ra.BPR(conf, sampler=sampler)
    model.fit(ds)
    h = model.history
    n_steps = (len(ds) + B - 1) // B
    t = _steady([e['train_time'] for e in h])
    loop_ms = t / n_steps * 1e3
    # the same steps issued back to back by a bare loop over the stepper fit drives (no loader, no dict, no loss list)
    stepper = model._fused_step.stepper if getattr(model, '_fused_step', None) is not None else None
    kernel_ms = None
    if stepper is not None:
        gen = torch.Generator(device=dev).manual_seed(5)
        K = min(n_steps, 200)
        uids = torch.randint(1, ds.num_users, (K + 1, B), device=dev, generator=gen)
        poss = torch.randint(1, ds.num_items, (K + 1, B), device=dev, generator=gen)

        def run(k):
            ticket = stepper.prepare(uids[0], poss[0])
            for i in range(1, k + 1):
                nxt = stepper.prepare(uids[i], poss[i])
                stepper.step(ticket)
                ticket = nxt
            stepper.step(ticket)
        run(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(K)
        torch.cuda.synchronize()
        kernel_ms = (time.perf_counter() - t0) / (K + 1) * 1e3
    out = {'B': B, 'steps_per_epoch': n_steps, 'epochs': len(h), 'train_s_per_epoch': round(t, 4),
           'loop_ms_per_step': round(loop_ms, 4), 'loop_M_triplets_s': round(B * n / loop_ms / 1e3, 1), 'learning_rate': lr,
           'train_loss_first_last': [round(h[0]['train_loss'], 4), round(h[-1]['train_loss'], 4)]}
    assert h[-1]['train_loss'] < h[0]['train_loss'] - 0.01, f"fit c2 B={B}: the training loss did not move: {out['train_loss_first_last']}" 
    if kernel_ms:
        out.update(stepper_ms_per_step=round(kernel_ms, 4), stepper_M_triplets_s=round(B * n / kernel_ms / 1e3, 1),
                   loop_over_stepper=round(kernel_ms / loop_ms, 3))
    del model
    torch.cuda.empty_cache()
    return out


def _ms(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def fit_c3(ra, dev, n_users=40_000, n_items=1_000_001, B=8192, n=256, d=128, L=50, epochs=3):
    """configs[2] through ``fit`` (the reference's loop: recommender.py:560-650; the tower: model/seq/sasrec.py:37-67) and the
    split of one step.  The pieces are timed one by one on ONE batch of the epoch's loader, each alone on the stream (their sum
    is close to, not equal to, the step: the loop also pays the loader and the host side)."""
    from recstudio_amd.retriever import _embedding_grad
    ds = synthetic_seq_dataset(ra, dev, n_users, n_items, max_seq_len=L)
    conf = {'model': {'embed_dim': d}, 'train': {'negative_count': n, 'batch_size': B, 'epochs': epochs, 'learning_rate': 1e-3,
                                                 'early_stop_patience': 10 ** 6}}
    model = ra.SASRec(conf, loss=ra.SampledSoftmaxLoss(), sampler=ra.PopularSamplerModel(ds.item_freq))
    t0 = time.perf_counter()
    model.fit(ds)
    wall = time.perf_counter() - t0
    h = model.history
    n_steps = (len(ds) + B - 1) // B
    t = _steady([e['train_time'] for e in h])
    out = {'workload': f'SASRec.fit (d={d}, max_seq_len={L}, 2 layers x 2 heads, dropout 0.5) + SampledSoftmaxLoss, popularity sampler '
                       f'n={n}, B={B}: synthetic sequences of {n_users} users over {n_items} items ({len(ds)} prefixes), device '
                       f'loader, torch Adam (dense, the reference\'s default)',
           'steps_per_epoch': n_steps, 'epochs': len(h), 'train_s_per_epoch': round(t, 3), 'ms_per_step': round(t / n_steps * 1e3, 3),
           'fit_wall_s': round(wall, 1), 'train_loss_first_last': [round(h[0]['train_loss'], 4), round(h[-1]['train_loss'], 4)]}
    assert h[-1]['train_loss'] < h[0]['train_loss'] - 0.01, f"fit c3: the training loss did not move: {out['train_loss_first_last']}"
    # ---- where a step goes: one full-width batch, the step's pieces one by one
    model.train()
    loader = ds.device_train_loader(B, shuffle=True, drop_last=True, device=dev)
    batch = None
    for b in loader:
        if batch is None or b['in_' + ds.fiid].shape[1] > batch['in_' + ds.fiid].shape[1]:
            batch = b                                   # the widest window of the epoch (L = max_seq_len)
    enc, iw = model.query_encoder, model.item_encoder.weight
    hist, (flat, start, end) = batch['in_' + ds.fiid], batch['_seg']
    Lb = hist.shape[1]
    pos = batch[ds.fiid]
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)

    def whole():
        opt.zero_grad()
        model.training_step(batch).backward()
        opt.step()
    parts = {'whole_step_one_batch': _ms(whole, 8, 3)}
    parts['history_gather_rsa_seg_gather'] = _ms(lambda: ra.ops.seg_gather(iw.detach(), flat, start, end, Lb, want_rows=True, want_ids=False), 20, 3)
    rows = ra.ops.seg_gather(iw.detach(), flat, start, end, Lb, want_rows=True, want_ids=False)[1]
    g_rows = torch.randn_like(rows)
    parts['history_scatter_backward'] = _ms(lambda: _embedding_grad(g_rows, hist, iw.shape[0]), 10, 2)
    # the tower alone: stock PyTorch (positions + 2 Transformer layers + last-position pooling), forward and backward
    tower_in = rows.detach().clone().requires_grad_(True)
    seqlen = batch['seqlen']

    def tower(x):
        positions = torch.arange(Lb, dtype=torch.long, device=dev).unsqueeze(0).expand(x.shape[0], Lb)
        seq = x + enc.position_emb(positions)
        causal = torch.triu(torch.ones(Lb, Lb, dtype=torch.bool, device=dev), 1)
        o = enc.transformer_layer(enc.dropout(seq), mask=causal, src_key_padding_mask=hist == 0)
        last = (seqlen - 1).clamp(min=0).view(-1, 1, 1).expand(-1, 1, o.shape[-1])
        return o.gather(1, last).squeeze(1)
    gq = torch.randn(hist.shape[0], d, device=dev)

    def tower_fb():
        tower_in.grad = None
        tower(tower_in).backward(gq)
    parts['transformer_fwd_bwd_stock_torch'] = _ms(tower_fb, 8, 3)
    with torch.no_grad():
        q = tower(rows).contiguous()
    kw = dict(pos_ids=pos, sampler=ra._native.SAMPLER_POPULAR, **model.sampler.lookup_kwargs())
    buf = {}

    def tail_fwd():        # sampling + negative / positive row gather + scores + SampledSoftmax + d loss/d query: ONE launch
        buf['o'] = ra.ops.fused_forward(iw.detach(), q, n, out=buf.get('o'), fused_loss='ssm', want_query_grad=True, **kw)
    parts['sample_score_loss_one_launch'] = _ms(tail_fwd, 20, 3)
    o = buf['o']
    one = torch.ones(1, device=dev)
    zero = {}

    def tail_bwd():        # dense weight.grad of the scored table: zero-fill + sorted, atomics-free scatter of every element
        zero['g'] = ra.ops.scatter_rows_sorted(torch.zeros_like(iw), q, o['neg_ids'], o['dneg'], pos_ids=pos, dpos=o['dpos'],
                                               upstream=one, pad_row=0)
    parts['item_gradient_scatter'] = _ms(tail_bwd, 10, 2)
    for p_ in model.parameters():
        if p_.grad is None:
            p_.grad = torch.zeros_like(p_)
    parts['optimizer_torch_adam_dense'] = _ms(opt.step, 10, 2)
    parts = {k: round(v, 3) for k, v in parts.items()}
    hot = (parts['history_gather_rsa_seg_gather'] + parts['history_scatter_backward'] + parts['sample_score_loss_one_launch'] +
           parts['item_gradient_scatter'])
    out['step_parts_ms'] = parts
    out['hot_path_ms'] = round(hot, 3)
    out['hot_path_share_of_step'] = round(hot / parts['whole_step_one_batch'], 4)
    out['step_parts_what'] = ('hot path = history gather + its backward scatter + the fused sample/score/loss launch + the item-'
                              'gradient scatter (in-tree HIP); the Transformer and the dense Adam update are stock PyTorch')
    del model, opt, zero, buf
    torch.cuda.empty_cache()
    return out


def fit_figures(ra, dev, n_items=10_000_001, n_users=1_000_001, n_inter=16_000_000, c1_epochs=10, batches=(65536, 4096)):
    out = {'c1_bpr_ml100k': fit_c1(ra, c1_epochs)}
    ds = synthetic_dataset(ra, dev, n_users, n_items, n_inter)
    out['c2_workload'] = (f'BPR.fit on a synthetic stream: {n_inter} interactions, {n_users} users, {n_items} items (Zipf(1), '
                          f'{int((ds.item_freq > 0).sum())} distinct items seen), d=128, popularity sampler n=64, '
                          f"device loader, train.fused_optimizer='sgd' one batch ahead")
    for B in batches:
        out[f'c2_B{B}'] = fit_c2(ra, dev, ds, B, epochs=4 if B >= 32768 else 3)
    del ds
    try:
        out['c3_sasrec'] = fit_c3(ra, dev)
    except AssertionError:
        raise
    except Exception as e:
        out['c3_sasrec'] = {'error': repr(e)[:300]}
    return out


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--items', type=int, default=10_000_001)
    ap.add_argument('--users', type=int, default=1_000_001)
    ap.add_argument('--inter', type=int, default=16_000_000)
    ap.add_argument('--only', default=None, choices=['c1', 'c2', 'c3'])
    ap.add_argument('--lr-per-sample', type=float, default=0.25)
    a = ap.parse_args()
    import recstudio_amd as ra
    dev = torch.device('cuda', 0)
    if a.only == 'c1':
        res = {'c1_bpr_ml100k': fit_c1(ra)}
    elif a.only == 'c2':
        ds = synthetic_dataset(ra, dev, a.users, a.items, a.inter)
        res = {f'c2_B{B}': fit_c2(ra, dev, ds, B, 3, lr_per_sample=a.lr_per_sample) for B in (65536, 4096)}
    elif a.only == 'c3':
        res = {'c3_sasrec': fit_c3(ra, dev)}
    else:
        res = fit_figures(ra, dev, a.items, a.users, a.inter)
    print(json.dumps(res))
