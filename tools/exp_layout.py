"""Which placement of the headline launch's nine output arrays is the slow one?  (tools/exp_outbuf.py: four big arrays anywhere
inside one arena with the five small ones elsewhere = 407 us; everything carved back to back at 4 KB = 445-465 us.)  One 1 GiB
arena, named layouts plus random ones (every array at a random 4 KiB-aligned offset), HIP events; one JSON line with the offsets
and the time of every layout -- for an offline fit of what correlates."""
import json
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recstudio_amd as ra                      # noqa: E402
from recstudio_amd import _native as nat        # noqa: E402
from bench import zipf_counts, prewarm          # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
N, U, d, B, n = 10_000_001, 1_000_001, 128, 65536, 64
NRAND = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def table(rows, seed):
    t = torch.empty(rows, d, device=dev).normal_(0, 0.02, generator=torch.Generator(device=dev).manual_seed(seed))
    t[0] = 0
    return t


def timed(fn, k=40, warm=0.25):
    prewarm(fn, warm)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / k * 1e3, 1)


item, user = table(N, 1), table(U, 2)
ps = ra.PopularSamplerModel(zipf_counts(N, 100_000_000)).to(dev)
gen = torch.Generator(device=dev).manual_seed(100)
uid = torch.randint(1, U, (B,), device=dev, generator=gen)
pos = torch.randint(1, N, (B,), device=dev, generator=gen)
kw = dict(sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs())
ARENA = 1 << 30
arena = torch.empty(ARENA, dtype=torch.uint8, device=dev)
other = torch.empty(64 << 20, dtype=torch.uint8, device=dev)       # a second allocation for "elsewhere"
BIG = [('neg_ids', torch.int64, B * n * 8), ('neg_score', torch.float32, B * n * 4), ('neg_logp', torch.float32, B * n * 4),
       ('dneg', torch.float32, B * n * 4)]
SMALL = [('pos_score', torch.float32, B * 4), ('pos_logp', torch.float32, B * 4), ('row_loss', torch.float32, B * 4),
         ('dpos', torch.float32, B * 4), ('loss', torch.float32, 4)]
SHAPE = {'neg_ids': (B, n), 'neg_score': (B, n), 'neg_logp': (B, n), 'dneg': (B, n), 'pos_score': (B,), 'pos_logp': (B,),
         'row_loss': (B,), 'dpos': (B,), 'loss': ()}


def build(offs, small_elsewhere=False):
    o = {}
    for name, dt, nb in BIG + SMALL:
        src = other if (small_elsewhere and nb <= B * 4) else arena
        o[name] = src[offs[name]:offs[name] + nb].view(dt).view(SHAPE[name])
    return o


def run(o):
    def step():
        ra.ops.fused_forward(item, user, n, out=o, fused_bpr=True, query_index=uid, pos_ids=pos, **kw)
    return timed(step)


def back_to_back(order, align):
    offs, t = {}, 0
    for name in order:
        nb = dict((a, c) for a, _, c in BIG + SMALL)[name]
        offs[name] = t
        t += (nb + align - 1) // align * align
    return offs


CARVE_ORDER = ['neg_ids', 'neg_score', 'pos_score', 'neg_logp', 'pos_logp', 'dneg', 'loss', 'row_loss', 'dpos']
BIG_FIRST = ['neg_ids', 'neg_score', 'neg_logp', 'dneg', 'pos_score', 'pos_logp', 'row_loss', 'dpos', 'loss']
res = {'named': {}, 'random': []}
small_offs = back_to_back([s[0] for s in SMALL], 4096)
for label, offs, elsewhere in (
        ('carve_order_4K', back_to_back(CARVE_ORDER, 4096), False),
        ('carve_order_2M', back_to_back(CARVE_ORDER, 2 << 20), False),
        ('big_first_4K', back_to_back(BIG_FIRST, 4096), False),
        ('big_first_2M', back_to_back(BIG_FIRST, 2 << 20), False),
        ('big_2M_small_elsewhere', dict(back_to_back([b[0] for b in BIG], 2 << 20), **small_offs), True),
        ('carve_order_4K_again', back_to_back(CARVE_ORDER, 4096), False)):
    res['named'][label] = run(build(offs, elsewhere))
rnd = random.Random(5)
sizes = dict((a, c) for a, _, c in BIG + SMALL)
for _ in range(NRAND):
    while True:
        offs = {name: rnd.randrange(0, (ARENA - sizes[name]) // 4096) * 4096 for name in sizes}
        iv = sorted((offs[k], offs[k] + sizes[k]) for k in sizes)
        if all(iv[i][1] <= iv[i + 1][0] for i in range(len(iv) - 1)):
            break
    res['random'].append({'offs': offs, 'us': run(build(offs))})
res['arena_ptr'] = hex(arena.data_ptr())
print(json.dumps(res))
