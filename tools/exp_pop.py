"""GPU box: fused forward (popular sampler + log-probs + fused BPR) vs guide_log2 at N = 1e7 and 1e8."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
from recstudio_amd import _native as nat
from bench import make_workload, zipf_counts, bytes_per_triplet
dev = torch.device('cuda', 0)
U, d, n, B = 1_000_001, 128, 64, 65536
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2]
for N, glogs in ((10_000_001, (23,)), (100_000_001, (25,))):
    item, user = make_workload(dev, N, U, d)
    counts = zipf_counts(N, 100_000_000)
    g = torch.Generator(device=dev).manual_seed(100)
    uid = torch.randint(1, U, (B,), device=dev, generator=g)
    pos = torch.randint(1, N, (B,), device=dev, generator=g)
    buf = {}
    def f_uni():
        buf['u'] = ra.ops.fused_forward(item, user, n, query_index=uid, pos_ids=pos, sampler=nat.SAMPLER_UNIFORM, out=buf.get('u'), fused_bpr=True)
    timeit(f_uni)
    print(f'N={N} uniform+bpr: {timeit(f_uni) * 1e3:.1f} us', flush=True)
    ids = torch.randint(1, N, (B, n), device=dev, generator=g)
    def f_given():
        buf['g'] = ra.ops.fused_forward(item, user, n, query_index=uid, pos_ids=pos, neg_ids=ids, out=buf.get('g'))
    print(f'N={N} given ids: {timeit(f_given) * 1e3:.1f} us', flush=True)
    for glog in glogs:
        ps = ra.PopularSamplerModel(counts, guide_log2=glog, lookup='lut').to(dev)
        key = f'p{glog}'
        def f_pop():
            buf[key] = ra.ops.fused_forward(item, user, n, query_index=uid, pos_ids=pos, sampler=nat.SAMPLER_POPULAR,
                                            table=ps.table, pop_prob=ps.pop_prob, guide=ps.guide, guide_log2=ps.guide_log2,
                                            out=buf.get(key), want_logp=True, table_prob=ps.table_prob, cdf_lut=ps.cdf_lut, fused_bpr=True)
        ms = timeit(f_pop)
        print(f'N={N} popular g={glog}: {ms * 1e3:.1f} us  {bytes_per_triplet(d, n, True) * B * n / ms / 1e6:.0f} GB/s', flush=True)
        def f_s():
            ra.ops.sample_popular(ps.table, ps.pop_prob, ps.guide, ps.guide_log2, B * n, cdf_lut=ps.cdf_lut)
        print(f'   sampler only: {timeit(f_s) * 1e3:.1f} us', flush=True)
        del ps
    del item, user
    torch.cuda.empty_cache()
