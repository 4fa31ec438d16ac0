"""GPU box: the stand-alone popularity sampler kernel (4.2 M ids, N = 1e7) with / without log-probs; run under
rocprofv3 --pmc FETCH_SIZE to get HBM bytes per sampled id."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
from recstudio_amd import _native as nat, rng
from recstudio_amd._native import ptr
from bench import zipf_counts
dev = torch.device('cuda', 0)
N, numel = 10_000_001, 65536 * 64
ps = ra.PopularSamplerModel(zipf_counts(N, 100_000_000)).to(dev)
ids = torch.empty(numel, dtype=torch.int64, device=dev)
logp = torch.empty(numel, dtype=torch.float32, device=dev)
u = torch.rand(numel, device=dev)
st = ra.ops._stream()
def run(kind):
    if kind == 'sample+logp':
        pc = rng.reserve(numel, 4, dev, None)
        nat.lib().rsa_sample_popular(ptr(ps.table), ptr(ps.pop_prob), ptr(ps.guide), N, ps.guide_log2, ptr(ids), ptr(logp), None, numel, pc.seed, pc.offset, pc.grid_threads, ptr(ps.cdf_lut), st)
    elif kind == 'sample':
        pc = rng.reserve(numel, 4, dev, None)
        nat.lib().rsa_sample_popular(ptr(ps.table), ptr(ps.pop_prob), ptr(ps.guide), N, ps.guide_log2, ptr(ids), None, None, numel, pc.seed, pc.offset, pc.grid_threads, ptr(ps.cdf_lut), st)
    elif kind == 'lookup+logp':
        nat.lib().rsa_popular_lookup(ptr(ps.table), ptr(ps.pop_prob), ptr(ps.guide), N, ps.guide_log2, ptr(u), ptr(ids), ptr(logp), numel, ptr(ps.cdf_lut), st)
    elif kind == 'nolut':
        nat.lib().rsa_popular_lookup(ptr(ps.table), ptr(ps.pop_prob), ptr(ps.guide), N, ps.guide_log2, ptr(u), ptr(ids), ptr(logp), numel, None, st)
for kind in ('sample+logp', 'sample', 'lookup+logp', 'nolut'):
    for _ in range(3): run(kind)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in evs:
        a.record(); run(kind); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    print(f'{kind:14s} {t[10] * 1e3:8.1f} us', flush=True)
