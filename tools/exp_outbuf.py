"""The same headline launch runs 404 us on one set of output buffers and 443 us on another (tools/exp_outliers.py: only the
addresses of neg_ids / neg_score / neg_logp / dneg differ).  Which relative placement of the four [B, n] output arrays is the
slow one?  One arena, the arrays at 64 MB strides plus a skew s per array (array k starts at k * 64 MB + k * s); then the whole
arena shifted.  `python tools/exp_outbuf.py` prints one JSON line."""
import json
import os
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


def table(rows, seed):
    t = torch.empty(rows, d, device=dev).normal_(0, 0.02, generator=torch.Generator(device=dev).manual_seed(seed))
    t[0] = 0
    return t


def timed(fn, k=60, warm=0.6):
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
first = ra.ops.fused_forward(item, user, n, fused_bpr=True, query_index=uid, pos_ids=pos, **kw)
arena = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
STRIDE = 64 << 20


def carve(off, shape, dtype):
    nb = torch.empty((), dtype=dtype).element_size()
    cnt = 1
    for s_ in shape:
        cnt *= s_
    return arena[off:off + cnt * nb].view(dtype).view(*shape)


def outputs(skew, shift=0):
    o = dict(first)                                   # the small ones ([B], scalars) stay where they are
    for k, (name, dt) in enumerate((('neg_ids', torch.int64), ('neg_score', torch.float32), ('neg_logp', torch.float32),
                                    ('dneg', torch.float32))):
        o[name] = carve(shift + k * STRIDE + k * skew, (B, n), dt)
    return o


def run(o):
    def step():
        ra.ops.fused_forward(item, user, n, out=o, fused_bpr=True, query_index=uid, pos_ids=pos, **kw)
    return timed(step)


res = {'default_allocations': run(first), 'ptrs': {k: hex(first[k].data_ptr()) for k in ('neg_ids', 'neg_score', 'neg_logp', 'dneg')}}
NAMES = (('neg_ids', torch.int64), ('neg_score', torch.float32), ('neg_logp', torch.float32), ('dneg', torch.float32))
# (b) separate allocations, each padded to P MB
res['separate_padded_MB'] = {}
for P in (34, 64, 128, 256, 512, 1024):
    o, keep = dict(first), []
    for name, dt in NAMES:
        raw = torch.empty(P << 20, dtype=torch.uint8, device=dev)
        keep.append(raw)
        nb = B * n * (8 if dt == torch.int64 else 4)
        o[name] = raw[:nb].view(dt).view(B, n)
    res['separate_padded_MB'][str(P)] = [run(o), hex(keep[1].data_ptr())]
    del o, keep
# (c) one arena of A MB, arrays back to back (2 MB aligned)
res['one_arena_MB'] = {}
for A in (86, 96, 128, 192, 256, 384, 512, 1024):
    ar = torch.empty(A << 20, dtype=torch.uint8, device=dev)
    o, off = dict(first), 0
    for name, dt in NAMES:
        nb = B * n * (8 if dt == torch.int64 else 4)
        o[name] = ar[off:off + nb].view(dt).view(B, n)
        off += (nb + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    res['one_arena_MB'][str(A)] = [run(o), hex(ar.data_ptr())]
    del o, ar
# (d) which of the four arrays matters: default allocations, ONE array moved into the big arena
res['one_moved_to_arena'] = {}
for k, (name, dt) in enumerate(NAMES):
    o = dict(first)
    o[name] = carve(k * STRIDE, (B, n), dt)
    res['one_moved_to_arena'][name] = run(o)
res['default_again'] = run(first)
print(json.dumps(res))
