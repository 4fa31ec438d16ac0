"""Bucket-line table size sweep at the bench shape (is a MALL-sized table worth the fallbacks?)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
from recstudio_amd import _native as nat
from bench import zipf_counts, bytes_per_triplet
dev = torch.device('cuda', 0)
d, B, n = 128, 65536, 64
def table(n_items, seed):
    t = torch.empty(n_items, d, device=dev).normal_(0, 0.02, generator=torch.Generator(device=dev).manual_seed(seed)); t[0] = 0; return t
def timeit(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs); return t[len(t) // 2] * 1e3
user = table(1_000_001, 3)
g = torch.Generator(device=dev).manual_seed(100)
for n_items, logs in ((10_000_001, (19, 20, 21, 22)), (100_000_001, (21, 22, 23))):
    item = table(n_items, 8)
    counts = zipf_counts(n_items, 100_000_000)
    uid = torch.randint(1, user.shape[0], (B,), device=dev, generator=g)
    pos = torch.randint(1, n_items, (B,), device=dev, generator=g)
    out = {}
    buf = {}
    def fu():
        buf['u'] = ra.ops.fused_forward(item, user, n, out=buf.get('u'), query_index=uid, pos_ids=pos, sampler=nat.SAMPLER_UNIFORM, fused_bpr=True, want_mean=False)
    out['uniform'] = round(timeit(fu), 1)
    for lg in logs:
        ps = ra.PopularSamplerModel(counts, lookup='lines', lines_log2=lg).to(dev)
        kw = ps.lookup_kwargs(); b2 = {}
        def f():
            b2['o'] = ra.ops.fused_forward(item, user, n, out=b2.get('o'), query_index=uid, pos_ids=pos, sampler=nat.SAMPLER_POPULAR, fused_bpr=True, want_mean=False, **kw)
        cnt = ps.cdf_lines[:, 0].view(torch.int32)
        out[f'g={lg}'] = {'us': round(timeit(f), 1), 'MB': ps.cdf_lines.numel() * 4 // 1000000, 'over8': round(float((cnt > 8).float().mean()), 4)}
        del ps, kw, b2
    out['uniform_again'] = round(timeit(fu), 1)
    print(n_items, json.dumps(out), flush=True)
    del item
