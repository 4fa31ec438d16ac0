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


def synthetic_dataset(ra, dev, n_users, n_items, n_inter, seed=1):
    """The configs[1] stream as a TripletDataset whose ids are already mapped (what ``load_cache`` yields): users uniform,
    items Zipf(1) over a permuted id space, drawn on the device; every interaction is a training sample."""
    g = torch.Generator(device=dev).manual_seed(seed)
    rank = torch.arange(1, n_items, dtype=torch.float64, device=dev)
    cdf = torch.cumsum(1.0 / rank, 0)
    cdf /= cdf[-1].clone()
    u = torch.rand(n_inter, dtype=torch.float64, device=dev, generator=g)
    by_rank = torch.searchsorted(cdf, u).clamp_(max=n_items - 2)
    perm = torch.randperm(n_items - 1, device=dev, generator=g)
    items = perm[by_rank] + 1
    users = torch.randint(1, n_users, (n_inter,), device=dev, generator=g)
    ds = ra.TripletDataset.from_mapped_ids(users.cpu(), items.cpu(), n_users=n_users, n_items=n_items, name='synthetic')
    del cdf, u, by_rank, perm, rank
    return ds


def fit_c2(ra, dev, ds, B, epochs, lr=0.05, d=128, n=64):
    sampler = ra.PopularSamplerModel(ds.item_freq)
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
           'loop_ms_per_step': round(loop_ms, 4), 'loop_M_triplets_s': round(B * n / loop_ms / 1e3, 1),
           'train_loss_first_last': [round(h[0]['train_loss'], 4), round(h[-1]['train_loss'], 4)]}
    if kernel_ms:
        out.update(stepper_ms_per_step=round(kernel_ms, 4), stepper_M_triplets_s=round(B * n / kernel_ms / 1e3, 1),
                   loop_over_stepper=round(kernel_ms / loop_ms, 3))
    del model
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
    return out


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--items', type=int, default=10_000_001)
    ap.add_argument('--users', type=int, default=1_000_001)
    ap.add_argument('--inter', type=int, default=16_000_000)
    ap.add_argument('--only', default=None, choices=['c1', 'c2'])
    a = ap.parse_args()
    import recstudio_amd as ra
    dev = torch.device('cuda', 0)
    if a.only == 'c1':
        res = {'c1_bpr_ml100k': fit_c1(ra)}
    elif a.only == 'c2':
        ds = synthetic_dataset(ra, dev, a.users, a.items, a.inter)
        res = {f'c2_B{B}': fit_c2(ra, dev, ds, B, 3) for B in (65536, 4096)}
    else:
        res = fit_figures(ra, dev, a.items, a.users, a.inter)
    print(json.dumps(res))
