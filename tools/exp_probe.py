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
lib = ctypes.CDLL(os.path.join(ROOT, 'tools', 'exp_tlb', 'libprobe.so'))
lib.probe_stream.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
lib.probe_tiles.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
T = B            # tiles of the headline launch; the pattern needs T * (512 + 3 * 256 + 16) B = 84.9 MB


def probe(kind, a):
    def fn():
        if kind == 'stream':
            lib.probe_stream(a.data_ptr(), 96 << 20, stream)
        elif kind == 'tiles':
            lib.probe_tiles(a.data_ptr(), T, None, 0, 0, 2048, stream)
        elif kind == 'tiles_small_grid':
            lib.probe_tiles(a.data_ptr(), T, None, 0, 0, 512, stream)
        else:
            lib.probe_tiles(a.data_ptr(), T, item.data_ptr(), N, 7, 2048, stream)
    return timed(fn, k=40, warm=0.1)


res = {'arena': []}
for a in arenas:
    rec = {'va': hex(a.data_ptr()), 'us': run(build(a, a))}
    for kind in ('stream', 'tiles', 'tiles_small_grid', 'tiles_reads'):
        rec[kind] = probe(kind, a)
    res['arena'].append(rec)
print(json.dumps(res))
