"""(variant: NAR separate 128 MB allocations instead of one arena)  Which placement of the headline launch's nine output arrays is the slow one?  (tools/exp_outbuf.py: four big arrays anywhere
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
ARENA = 128 << 20
NAR = int(sys.argv[1]) if len(sys.argv) > 1 else 12
arenas = [torch.empty(ARENA, dtype=torch.uint8, device=dev) for _ in range(NAR)]      # all alive: NAR distinct allocations
BIG = [('neg_ids', torch.int64, B * n * 8), ('neg_score', torch.float32, B * n * 4), ('neg_logp', torch.float32, B * n * 4),
       ('dneg', torch.float32, B * n * 4)]
SMALL = [('pos_score', torch.float32, B * 4), ('pos_logp', torch.float32, B * 4), ('row_loss', torch.float32, B * 4),
         ('dpos', torch.float32, B * 4), ('loss', torch.float32, 4)]
SHAPE = {'neg_ids': (B, n), 'neg_score': (B, n), 'neg_logp': (B, n), 'dneg': (B, n), 'pos_score': (B,), 'pos_logp': (B,),
         'row_loss': (B,), 'dpos': (B,), 'loss': ()}
offs, t = {}, 0
for name, _, nb in BIG + SMALL:
    offs[name] = t
    t += (nb + 4095) // 4096 * 4096


def build(big_arena, small_arena):
    o = {}
    for name, dt, nb in BIG + SMALL:
        src = small_arena if nb <= B * 4 else big_arena
        o[name] = src[offs[name]:offs[name] + nb].view(dt).view(SHAPE[name])
    return o


def run(o):
    def step():
        ra.ops.fused_forward(item, user, n, out=o, fused_bpr=True, query_index=uid, pos_ids=pos, **kw)
    return timed(step)


import ctypes                                   # noqa: E402
from recstudio_amd import placement             # noqa: E402
lib = nat.lib()
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def probe_with(a, src):
    def fn():
        lib.rsa_placement_probe(ctypes.c_void_p(a.data_ptr()), a.numel(), ctypes.c_void_p(src.data_ptr()), src.numel() * src.element_size(), 7, stream)
    return timed(fn, k=20, warm=0.05)


res = {'arena': []}
for a in arenas:
    res['arena'].append({'va': hex(a.data_ptr()), 'probe': round(placement.probe_us(a), 1)})
order = sorted(range(len(arenas)), key=lambda i: res['arena'][i]['probe'])
f1, f2, s1, s2 = order[0], order[1], order[-1], order[-2]
res['fast_slow'] = [f1, f2, s1, s2]


def ids_in(a):
    """uid / pos copied into arena `a` (behind where the outputs would be: the last 2 MB)"""
    base = a.numel() - (2 << 20)
    u = a[base:base + B * 8].view(torch.int64)
    p = a[base + (1 << 20):base + (1 << 20) + B * 8].view(torch.int64)
    u.copy_(uid)
    p.copy_(pos)
    return u, p


def run_with(out_arena, id_arena):
    o = build(out_arena, out_arena)
    u, p = ids_in(id_arena)
    kw2 = dict(kw)

    def step():
        ra.ops.fused_forward(item, user, n, out=o, fused_bpr=True, query_index=u, pos_ids=p, **kw2)
    return timed(step)


res['out_fast_ids_original'] = run(build(arenas[f1], arenas[f1]))
res['out_fast_ids_fast'] = run_with(arenas[f1], arenas[f2])
res['out_fast_ids_slow'] = run_with(arenas[f1], arenas[s1])
res['out_slow_ids_fast'] = run_with(arenas[s1], arenas[f2])
res['out_slow_ids_slow'] = run_with(arenas[s1], arenas[s2])
# the user table (512 MB, read: one row per query) copied into four fast / four slow arenas is too big for one arena; the
# sampler's small tables instead: pop_prob / table views are left alone
print(json.dumps(res))
