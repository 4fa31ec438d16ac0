"""Experiment driver (GPU box): time variants of the fused forward at the bench size."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
from recstudio_amd import _native as nat
from bench import make_workload, zipf_counts, bytes_per_triplet

dev = torch.device('cuda', 0)
N, U, d, n = 10_000_001, 1_000_001, 128, 64
B = int(os.environ.get('B', 65536))
item, user = make_workload(dev, N, U, d)
counts = zipf_counts(N, 100_000_000)
g = torch.Generator(device=dev).manual_seed(100)
uid = torch.randint(1, U, (B,), device=dev, generator=g)
pos = torch.randint(1, N, (B,), device=dev, generator=g)

def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t)//2], t[0]

res = {}
def report(name, ms, alg):
    res[name] = (round(ms[0]*1e3,1), round(alg/ms[0]/1e6,1))
    print(f'{name:40s} median {ms[0]*1e3:8.1f} us  min {ms[1]*1e3:8.1f} us  {alg/ms[0]/1e6:8.1f} GB/s alg  {B*n/ms[0]/1e3:8.1f} M trip/s', flush=True)

# given ids (uniform random) : pure gather+score
ids = torch.randint(1, N, (B, n), device=dev, generator=g)
buf = {}
def f_given():
    buf['g'] = ra.ops.fused_forward(item, user, n, query_index=uid, pos_ids=pos, neg_ids=ids, out=buf.get('g'))
report('given ids (uniform random)', timeit(f_given), (4*d + 8*d/n + 16/n + 8 + 4) * B * n)
us = ra.UniformSampler(N)
def f_uni():
    buf['u'] = ra.ops.fused_forward(item, user, n, query_index=uid, pos_ids=pos, sampler=nat.SAMPLER_UNIFORM, out=buf.get('u'))
report('uniform sampler', timeit(f_uni), bytes_per_triplet(d, n, False) * B * n)
for glog in (22, 23, 24):
    ps = ra.PopularSamplerModel(counts, guide_log2=glog, lookup='lut').to(dev)
    for logp, pairs, lut in ((True, True, False), (True, True, True), (True, False, True), (False, True, True)):
        key = f'p{glog}{logp}{pairs}{lut}'
        def f_pop():
            o = buf.get(key)
            buf[key] = ra.ops.fused_forward(item, user, n, query_index=uid, pos_ids=pos, sampler=nat.SAMPLER_POPULAR,
                                            table=ps.table, pop_prob=ps.pop_prob, guide=ps.guide, guide_log2=ps.guide_log2,
                                            out=o, want_logp=logp, table_prob=ps.table_prob if pairs else None,
                                            cdf_lut=ps.cdf_lut if lut else None)
        report(f'popular guide_log2={glog} logp={logp} pairs={pairs} lut={lut}', timeit(f_pop), bytes_per_triplet(d, n, True) * B * n)
    # stand-alone sampler kernel
    def f_s():
        ra.ops.sample_popular(ps.table, ps.pop_prob, ps.guide, ps.guide_log2, B * n)
    report(f'  sampler only guide_log2={glog}', timeit(f_s), 16 * B * n)
print(json.dumps(res))
