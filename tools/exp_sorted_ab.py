"""In-process A/B of the sorted apply pass (rsa_rows_update_presorted) between library builds: the step's pairs are sorted
once, then alternating rounds of launches of the apply pass through each library over the same workspace.
SHAPE=all (headline shape, every element applied) | presorted (solo rows skipped, as after the in-forward update) |
dq / item (the two sorted scatters of the sharded backward on the owner: 4.2 M elements onto 4096 query rows / onto their item rows)
usage: SHAPE=all python tools/exp_sorted_ab.py name=lib.so ..."""
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

shape = os.environ.get('SHAPE', 'all')
dev = torch.device('cuda', 0)
d = 128
gen = torch.Generator(device=dev).manual_seed(5)
if shape in ('dq', 'item'):
    # the sharded backward on the owner (shard.HipBackend.backward_segments): 4.2 M received elements (row, query index, d)
    nq, G_B, M, n = 12_500_001, 4096, 4096 * 1025, 1
    rows = torch.randint(1, nq, (M,), device=dev, generator=gen)
    qsel = torch.arange(G_B, device=dev).repeat_interleave(1025)[torch.randperm(M, device=dev, generator=gen)]
    pos = None
    if shape == 'dq':        # qgrad_all[qidx] += d * item_local[row]: keys = query indices (runs of 1025), reads = item rows
        N, query, qidx, neg = G_B, torch.empty(nq, d, device=dev).normal_(0, 0.02, generator=gen), rows, qsel.view(M, 1).contiguous()
    else:                    # item_grad_local[row] += d * q_all[qidx]: keys = item rows (mostly alone), reads = query rows
        N, query, qidx, neg = nq, torch.empty(G_B, d, device=dev).normal_(0, 0.02, generator=gen), qsel, rows.view(M, 1).contiguous()
else:
    N, M, n = 10_000_001, 65536, 64
    query = torch.empty(1_000_001, d, device=dev).normal_(0, 0.02, generator=gen)
    qidx = torch.randint(1, 1_000_001, (M,), device=dev, generator=gen)
    ps = ra.PopularSamplerModel(zipf_counts(N, 100_000_000)).to(dev)
    pos = torch.randint(1, N, (M,), device=dev, generator=gen)
    neg = ps(torch.empty(M, 1, device=dev), n, None)[0]
target = torch.zeros(N, d, device=dev)
dneg = torch.randn(M, n, device=dev, generator=gen) * 1e-3
dpos = torch.randn(M, device=dev, generator=gen) * 1e-3 if pos is not None else None
solo, ws = ra.ops.sort_step_elements(pos, neg, N, pad_row=0, want_solo=(shape == 'presorted'))
up = torch.full((1,), -1e-3, device=dev)
libs = {'default': nat.lib()}
for spec in sys.argv[1:]:
    name, path = spec.split('=')
    h = ctypes.CDLL(path)
    h.rsa_rows_update_presorted.restype, h.rsa_rows_update_presorted.argtypes = nat.SIGNATURES['rsa_rows_update_presorted']
    libs[name] = h
stream = ra.ops._stream()


ADAM = os.environ.get('ADAM') == '1'
if ADAM:
    m_state, v_state = torch.zeros_like(target), torch.zeros_like(target)


def launch(h):
    a = nat.RowsUpdateArgs()
    a.query, a.query_index, a.n_query_rows, a.dim, a.has_pos = ptr(query), ptr(qidx), query.shape[0], d, int(pos is not None)
    a.n_queries, a.num_neg, a.dpos, a.dneg, a.n_items, a.pad_row, a.target = M, n, ptr(dpos), ptr(dneg), N, 0, ptr(target)
    a.workspace, a.workspace_bytes = ptr(ws), ws.numel()
    if ADAM:
        a.exp_avg, a.exp_avg_sq, a.lr, a.beta1, a.beta2, a.eps, a.step = ptr(m_state), ptr(v_state), 1e-3, 0.9, 0.999, 1e-8, 3
    else:
        a.upstream = ptr(up)
    rc = h.rsa_rows_update_presorted(ctypes.byref(a), stream)
    assert rc == 0


res = {k: [] for k in libs}
for h in libs.values():
    for _ in range(5):
        launch(h)
torch.cuda.synchronize()
for rnd in range(5):
    for name, h in libs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            launch(h)
        e1.record()
        torch.cuda.synchronize()
        res[name].append(round(e0.elapsed_time(e1) / 20 * 1e3, 1))
print(json.dumps({'shape': shape, 'us_per_launch': {k: [min(v), sorted(v)[len(v) // 2]] for k, v in res.items()}}))
