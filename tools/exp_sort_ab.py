"""In-process A/B of the step's sort (rsa_sort_step_elements: in-tree radix sort + solo classification) between library
builds, at several element counts around the 1024-workgroup boundary of the scatter pass (4 workgroups of 39 KB LDS per CU x
256 CUs): alternating rounds of launches through each library over the same ids, HIP events around each round.
usage: python tools/exp_sort_ab.py [name=lib.so ...]        (env SIZES="65536x64 64512x64 4096x1024", SOLO=0/1, N=10000001)"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recstudio_amd as ra                      # noqa: E402
from recstudio_amd import _native as nat        # noqa: E402
from recstudio_amd._native import ptr           # noqa: E402
from bench import zipf_counts                   # noqa: E402

dev = torch.device('cuda', 0)
N = int(os.environ.get('N', 10_000_001))
SOLO = os.environ.get('SOLO', '1') == '1'
sizes = [tuple(int(v) for v in s.split('x')) for s in os.environ.get('SIZES', '65536x64 64512x64 63000x64 4096x1024').split()]
gen = torch.Generator(device=dev).manual_seed(5)
libs = {'default': nat.lib()}
for spec in sys.argv[1:]:
    name, path = spec.split('=')
    h = ctypes.CDLL(path)
    for fn in ('rsa_sort_step_elements', 'rsa_scatter_rows_sorted_workspace_bytes'):
        getattr(h, fn).restype, getattr(h, fn).argtypes = nat.SIGNATURES[fn]
    libs[name] = h
stream = ra.ops._stream()
ps = ra.PopularSamplerModel(zipf_counts(N, 100_000_000)).to(dev)
out = {}
for M, n in sizes:
    pos = torch.randint(1, N, (M,), device=dev, generator=gen)
    neg = ps(torch.empty(M, 1, device=dev), n, None)[0]
    solo = torch.empty(M, n + 1, dtype=torch.uint8, device=dev) if SOLO else None
    res, keep = {}, {}
    for name, h in libs.items():
        wsb = int(h.rsa_scatter_rows_sorted_workspace_bytes(M, n, N))
        keep[name] = (torch.empty(wsb, dtype=torch.uint8, device=dev), wsb)
    def run(name, reps):
        h, (ws, wsb) = libs[name], keep[name]
        for _ in range(reps):
            a = nat.RowsUpdateArgs()
            a.pos_ids, a.neg_ids, a.n_queries, a.num_neg, a.n_items, a.pad_row = ptr(pos), ptr(neg), M, n, N, 0
            a.solo, a.workspace, a.workspace_bytes = ptr(solo), ptr(ws), wsb
            rc = h.rsa_sort_step_elements(ctypes.byref(a), stream)
            assert rc == 0, rc
    for name in libs:
        run(name, 20)
    torch.cuda.synchronize()
    for rnd in range(5):
        for name in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(name, 40)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 40 * 1e3)
    # every variant must leave the stable sort of (id, element) in one of the two pair buffers
    tot = M * (n + 1)
    ids = torch.cat([pos.view(M, 1), neg.view(M, n)], 1).reshape(-1)
    want = (ids << 32 | torch.arange(tot, device=dev)).sort().values
    for name in libs:
        keep[name][0].zero_()
        run(name, 1)
        torch.cuda.synchronize()
        ws = keep[name][0]
        seg = (tot * 8 + 255) // 256 * 256
        a, b = ws[:tot * 8].view(torch.int64), ws[seg:seg + tot * 8].view(torch.int64)
        mask = ~(1 << 31) if SOLO else -1           # the classification flags solo elements in the payload's top bit
        ok = bool(((a & mask) == want).all()) or bool(((b & mask) == want).all())
        assert ok, (name, int(((a & mask) != want).sum()), int(((b & mask) != want).sum()))
    out[f'{M}x{n}'] = {k: round(min(v), 1) for k, v in res.items()}
    out[f'{M}x{n}']['tiles_4096'] = (M * (n + 1) + 4095) // 4096
print(json.dumps(out))
