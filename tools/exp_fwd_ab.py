"""In-process A/B of the fused forward kernels between library builds: ONE set of tables and buffers, the call frozen by
ops.FusedStep, alternating rounds of 50 launches through each library's rsa_fused_sample_gather_score (process-to-process
placement noise is +-4 %; in one process the rounds repeat to +-0.2 %).
usage: SHAPE=headline_pop|headline_uni|given|n1e8_uni|n1e8_pop|b4096|b16384|train|ssm|ssm_train|walk|walk_train|upd python tools/exp_fwd_ab.py name=lib.so ..."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recstudio_amd as ra                      # noqa: E402
from recstudio_amd import _native as nat        # noqa: E402
from bench import zipf_counts                   # noqa: E402

shape = os.environ.get('SHAPE', 'headline_pop')
dev = torch.device('cuda', 0)
d, U = 128, 1_000_001
gen = torch.Generator(device=dev).manual_seed(100)


def table(n_items):
    t = torch.empty(n_items, d, device=dev).normal_(0, 0.02, generator=gen)
    t[0] = 0
    return t


user = torch.empty(U, d, device=dev).normal_(0, 0.02, generator=gen)
N = 10_000_001
n, B = 64, 65536
kw = dict(fused_bpr=True)
if shape in ('n1e8_uni', 'n1e8_pop'):
    N = 100_000_001
    if shape == 'n1e8_pop':
        ps_path = '/tmp/rsa_ps_1e8.pt'
        if os.path.exists(ps_path):
            ps = torch.load(ps_path, weights_only=False).to(dev)
        else:
            ps = ra.PopularSamplerModel(zipf_counts(N, 100_000_000))
            torch.save(ps, ps_path)
            ps = ps.to(dev)
        kw.update(sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs())
    else:
        kw.update(sampler=nat.SAMPLER_UNIFORM)
elif shape in ('headline_pop', 'b4096', 'b16384', 'train', 'upd'):
    B = {'b4096': 4096, 'b16384': 16384}.get(shape, 65536)
    ps_path = '/tmp/rsa_ps_1e7.pt'
    if os.path.exists(ps_path):
        ps = torch.load(ps_path, weights_only=False).to(dev)
    else:
        ps = ra.PopularSamplerModel(zipf_counts(N, 100_000_000))
        torch.save(ps, ps_path)
        ps = ps.to(dev)
    kw.update(sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs())
elif shape == 'headline_uni':
    kw.update(sampler=nat.SAMPLER_UNIFORM)
elif shape in ('ssm', 'ssm_train'):
    N, n, B = 1_000_001, 256, 8192
    ps = ra.PopularSamplerModel(zipf_counts(N, 100_000_000)).to(dev)
    kw = dict(fused_loss='ssm', sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs())
elif shape in ('walk', 'walk_train', 'walk_1e8'):
    N, n, B = (100_000_001 if shape == 'walk_1e8' else 12_500_001), 1024, 4096
    kw.update(sampler=nat.SAMPLER_UNIFORM, want_mean=False)
item = table(N)
uid = torch.randint(1, U, (B,), device=dev, generator=gen)
pos = torch.randint(1, N, (B,), device=dev, generator=gen)
if shape == 'given':
    kw.update(sampler=nat.SAMPLER_GIVEN, neg_ids=torch.randint(1, N, (B, n), device=dev, generator=gen))
if shape in ('train', 'ssm_train', 'walk_train'):
    kw.update(want_query_grad=True)
if shape == 'upd':
    neg = ps(torch.empty(B, 1, device=dev), n, None)[0]
    solo, ws = ra.ops.sort_step_elements(pos, neg, N, pad_row=0)
    step_scale = torch.full((1,), -1e-6, device=dev)
    for k in ('table', 'pop_prob', 'guide', 'guide_log2', 'table_prob', 'cdf_lut', 'cdf_lines', 'lines_log2'):
        kw.pop(k, None)
    kw.update(sampler=nat.SAMPLER_GIVEN, neg_ids=neg, want_query_grad=True, inplace_update=(solo, step_scale))
if shape in ('ssm', 'ssm_train'):
    fs = ra.ops.FusedStep(item, user[1:B + 1].contiguous(), n, pos_ids=pos, **kw)
else:
    fs = ra.ops.FusedStep(item, user, n, query_index=uid, pos_ids=pos, **kw)
fns = {'default': fs._fn}
for spec in sys.argv[1:]:
    name, path = spec.split('=')
    fn = ctypes.CDLL(path).rsa_fused_sample_gather_score
    fn.restype, fn.argtypes = nat.SIGNATURES['rsa_fused_sample_gather_score']
    fns[name] = fn
res = {k: [] for k in fns}
for fn in fns.values():
    fs._fn = fn
    for _ in range(20):
        fs()
torch.cuda.synchronize()
for rnd in range(5):
    for name, fn in fns.items():
        fs._fn = fn
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fs()
        e1.record()
        torch.cuda.synchronize()
        res[name].append(round(e0.elapsed_time(e1) / 50 * 1e3, 1))
print(json.dumps({'shape': shape, 'B': B, 'n': n, 'us_per_step': {k: [min(v), sorted(v)[len(v) // 2]] for k, v in res.items()}}))
