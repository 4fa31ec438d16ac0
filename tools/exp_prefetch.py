"""In-place SGD step at the headline shape: plain bpr_sgd_step vs fused.PrefetchedBPRSGD (next step's sampling + sort on a
side stream).  usage: python tools/exp_prefetch.py [popular|uniform]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recstudio_amd as ra                      # noqa: E402
from bench import zipf_counts                   # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else 'popular'
dev = torch.device('cuda', 0)
N, U, d, B, n = 10_000_001, 1_000_001, 128, 65536, 64
item = torch.empty(N, d, device=dev).normal_(0, 0.02)
item[0] = 0
user = torch.empty(U, d, device=dev).normal_(0, 0.02)
gen = torch.Generator(device=dev).manual_seed(100)
uid = torch.randint(1, U, (B,), device=dev, generator=gen)
pos = torch.randint(1, N, (B,), device=dev, generator=gen)
sampler = (ra.PopularSamplerModel(zipf_counts(N, 100_000_000)) if kind == 'popular' else ra.UniformSampler(N)).to(dev)


def timed(fn, K=100, W=20):
    for _ in range(W):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


res = {'sampler': kind}
for rep in range(3):
    res.setdefault('plain_ms', []).append(round(timed(lambda: ra.fused.bpr_sgd_step(item, user, n, 1e-3, user_ids=uid, pos_ids=pos, sampler=sampler)), 4))
    st = ra.fused.PrefetchedBPRSGD(item, user, n, 1e-3, sampler)
    box = {'t': st.prepare(uid, pos)}

    def pf():
        nxt = st.prepare(uid, pos)
        st.step(box['t'])
        box['t'] = nxt
    res.setdefault('prefetched_ms', []).append(round(timed(pf), 4))
print(json.dumps(res))
# lazy Adam: plain step vs the prepare / step_prepared pair one batch ahead
fa = ra.fused.FusedBPRAdam(item, user, lr=1e-3)
for rep in range(3):
    res.setdefault('adam_plain_ms', []).append(round(timed(lambda: fa.step(n, user_ids=uid, pos_ids=pos, sampler=sampler), 40, 5), 4))
    abox = {'t': fa.prepare(n, user_ids=uid, pos_ids=pos, sampler=sampler)}

    def apf():
        nxt = fa.prepare(n, user_ids=uid, pos_ids=pos, sampler=sampler)
        fa.step_prepared(abox['t'])
        abox['t'] = nxt
    res.setdefault('adam_prefetched_ms', []).append(round(timed(apf, 40, 5), 4))
print(json.dumps(res))
