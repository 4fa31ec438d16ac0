"""Experiment (small batches): the forward step with its negatives drawn ONE STEP AHEAD by the stand-alone
sampler kernel on a second stream (rsa_sample_popular: the same ids and log-probabilities the in-kernel sampler would draw,
same generator consumption), the scoring launch reading them as given ids -- against the single fused launch.
usage: B=4096 python tools/exp_presample.py"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recstudio_amd as ra                      # noqa: E402
from recstudio_amd import _native as nat        # noqa: E402
from recstudio_amd import rng                   # noqa: E402
from recstudio_amd._native import ptr           # noqa: E402
from bench import zipf_counts                   # noqa: E402

dev = torch.device('cuda', 0)
d, U, N, n = 128, 1_000_001, 10_000_001, 64
B = int(os.environ.get('B', 4096))
g = torch.Generator(device=dev).manual_seed(1)
item = torch.empty(N, d, device=dev).normal_(0, 0.02, generator=g)
user = torch.empty(U, d, device=dev).normal_(0, 0.02, generator=g)
ps = ra.PopularSamplerModel(zipf_counts(N, 100_000_000)).to(dev)
kw = ps.lookup_kwargs()
uid = torch.randint(1, U, (B,), device=dev, generator=g)
pos = torch.randint(1, N, (B,), device=dev, generator=g)
lib = nat.lib()

# ---- A: the single fused launch (what bench.py's sweep times)
fused = ra.ops.FusedStep(item, user, n, fused_bpr=True, query_index=uid, pos_ids=pos, sampler=nat.SAMPLER_POPULAR, **kw)

# ---- B: sampler kernel one step ahead on a second stream + scoring launch over given ids
side = torch.cuda.Stream()
bufs = []
for _ in range(2):
    ids = torch.empty(B, n, dtype=torch.int64, device=dev)
    logp = torch.empty(B, n, dtype=torch.float32, device=dev)
    ids.fill_(1)
    step = ra.ops.FusedStep(item, user, n, fused_bpr=True, query_index=uid, pos_ids=pos, sampler=nat.SAMPLER_GIVEN, neg_ids=ids)
    bufs.append({'ids': ids, 'logp': logp, 'step': step, 'sampled': torch.cuda.Event(), 'scored': torch.cuda.Event()})
table, prob = kw['table'], kw['pop_prob']
lines, lg = kw.get('cdf_lines'), int(kw.get('lines_log2') or 0)


def sample_into(b):
    pc = rng.reserve(B * n, 4, dev, None)
    rc = lib.rsa_sample_popular(ptr(table), ptr(prob), None, table.numel(), 0, ptr(b['ids']), ptr(b['logp']), None, B * n, pc.seed,
                                pc.offset, pc.grid_threads, pc.elem_base, None, ptr(lines), lg,
                                ctypes.c_void_p(side.cuda_stream))
    nat.check(rc, "rsa_sample_popular")


main = torch.cuda.current_stream()
state = {'k': 0}
with torch.cuda.stream(side):
    sample_into(bufs[0])
    bufs[0]['sampled'].record(side)
for b in bufs:
    b['scored'].record(main)


def step_b():
    k = state['k']
    cur, nxt = bufs[k & 1], bufs[(k + 1) & 1]
    side.wait_event(nxt['scored'])              # the launch that read this buffer two steps ago is done
    sample_into(nxt)                            # step k + 1's negatives, under step k's scoring launch
    nxt['sampled'].record(side)
    main.wait_event(cur['sampled'])
    cur['step']()
    cur['scored'].record(main)
    state['k'] = k + 1


def timed(fn, K=200, W=30):
    for _ in range(W):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / K * 1e3, 2)


res = {'B': B, 'fused_us': [], 'presampled_us': []}
for r in range(3):
    res['fused_us'].append(timed(fused))
    res['presampled_us'].append(timed(step_b))
# same draws: one fused step and one presampled step from the same generator state
st = torch.cuda.get_rng_state(dev)
out_a = fused()
ids_a = out_a['neg_ids'].clone()
torch.cuda.set_rng_state(st, dev)
with torch.cuda.stream(side):
    sample_into(bufs[0])
torch.cuda.synchronize()
res['same_ids'] = bool(torch.equal(ids_a, bufs[0]['ids']))
alg = (4 * d + 2 * 4 * d / n + 16.0 / n + 16 + 8 + 4 + 8.0 / n) * B * n
res['frac_fused'] = round(alg / min(res['fused_us']) / 8e6, 4)
res['frac_presampled'] = round(alg / min(res['presampled_us']) / 8e6, 4)
print(json.dumps(res))
