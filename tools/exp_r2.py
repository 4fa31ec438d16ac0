"""Round-2 experiment driver (GPU box): popularity lookup forms (LUT vs bucket lines) at N = 1e7 / 1e8, the fused
SampledSoftmax step, small batches.  RSA_LIB selects a variant build (tools/build_variant.sh)."""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
from recstudio_amd import _native as nat
from bench import zipf_counts, bytes_per_triplet

dev = torch.device('cuda', 0)
d = 128
what = sys.argv[1:] or ['pop7', 'pop8', 'ssm', 'small']
res = {'lib': os.path.basename(nat.LIB_PATH)}


def timeit(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2] * 1e3      # us


def table(n_items, seed):
    t = torch.empty(n_items, d, device=dev).normal_(0, 0.02, generator=torch.Generator(device=dev).manual_seed(seed))
    t[0] = 0
    return t


user = table(1_000_001, 3)
g = torch.Generator(device=dev).manual_seed(100)


def pop_bench(tag, n_items, B=65536, n=64):
    item = table(n_items, 8)
    counts = zipf_counts(n_items, 100_000_000)
    uid = torch.randint(1, user.shape[0], (B,), device=dev, generator=g)
    pos = torch.randint(1, n_items, (B,), device=dev, generator=g)
    alg = bytes_per_triplet(d, n, True, fused_loss=True) * B * n
    out = {}
    for lookup in ('lut', 'lines'):
        ps = ra.PopularSamplerModel(counts, lookup=lookup).to(dev)
        kw = ps.lookup_kwargs()
        buf = {}
        for name, extra in (('fwd_bpr', dict(fused_bpr=True, want_mean=False)),
                            ('fwd_bpr_qg', dict(fused_bpr=True, want_mean=False, want_query_grad=True))):
            def f():
                buf[name] = ra.ops.fused_forward(item, user, n, out=buf.get(name), query_index=uid, pos_ids=pos,
                                                 sampler=nat.SAMPLER_POPULAR, **kw, **extra)
            us = timeit(f)
            out[f'{lookup}/{name}'] = {'us': round(us, 1), 'frac': round(alg / us / 1e3 / 8000, 4)}
        out[f'{lookup}/meta'] = {'guide_log2': ps.guide_log2, 'lines_log2': ps.lines_log2}
        def fs():
            ra.ops.sample_popular(ps.table, ps.pop_prob, ps.guide, ps.guide_log2, B * n, cdf_lut=ps.cdf_lut,
                                  cdf_lines=ps.cdf_lines, lines_log2=ps.lines_log2)
        out[f'{lookup}/sampler_only_us'] = round(timeit(fs), 1)
        del ps, kw, buf
        torch.cuda.empty_cache()
    buf = {}
    def fu():
        buf['u'] = ra.ops.fused_forward(item, user, n, out=buf.get('u'), query_index=uid, pos_ids=pos,
                                        sampler=nat.SAMPLER_UNIFORM, fused_bpr=True, want_mean=False)
    us = timeit(fu)
    out['uniform/fwd_bpr'] = {'us': round(us, 1), 'frac': round(bytes_per_triplet(d, n, False, fused_loss=True) * B * n / us / 1e3 / 8000, 4)}
    res[tag] = out
    print(tag, json.dumps(out), flush=True)


if 'pop7' in what:
    pop_bench('pop_N1e7', 10_000_001)
if 'pop8' in what:
    pop_bench('pop_N1e8', 100_000_001)

if 'ssm' in what:
    n_items, B, n = 1_000_001, 8192, 256
    item = table(n_items, 8)
    counts = zipf_counts(n_items, 100_000_000)
    ps = ra.PopularSamplerModel(counts).to(dev)
    q = user[1:B + 1].contiguous()
    pos = torch.randint(1, n_items, (B,), device=dev, generator=g)
    kw = dict(pos_ids=pos, sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs())
    buf = {}
    def unfused():
        buf['a'] = ra.ops.fused_forward(item, q, n, out=buf.get('a'), **kw)
        o = buf['a']
        return ra.ops.pairwise_loss(nat.LOSS_SSM, o['pos_score'], o['neg_score'], o['pos_logp'], o['neg_logp'])
    def unfused_train():
        loss, dpos, dneg, _ = unfused()
        o = buf['a']
        return ra.ops.fused_backward(item, q, o['neg_ids'], dneg, pos_ids=pos, dpos=dpos, dense_item_grad=False,
                                     row_item_grad=True, want_query_grad=True)
    def fused():
        buf['b'] = ra.ops.fused_forward(item, q, n, out=buf.get('b'), fused_loss='ssm', **kw)
    def fused_qg():
        buf['c'] = ra.ops.fused_forward(item, q, n, out=buf.get('c'), fused_loss='ssm', want_query_grad=True, **kw)
    def fused_train():
        fused_qg()
        o = buf['c']
        return ra.ops.fused_backward(item, q, o['neg_ids'], o['dneg'], pos_ids=pos, dpos=o['dpos'], dense_item_grad=False,
                                     row_item_grad=True, want_query_grad=False)
    alg = bytes_per_triplet(d, n, True) * B * n
    out = {}
    for name, f in (('unfused_fwd_loss', unfused), ('fused_fwd_loss', fused), ('fused_fwd_loss_qg', fused_qg),
                    ('unfused_train', unfused_train), ('fused_train', fused_train)):
        us = timeit(f)
        out[name] = {'us': round(us, 1), 'fwd_alg_frac': round(alg / us / 1e3 / 8000, 4)}
    res['ssm_B8192_n256'] = out
    print('ssm', json.dumps(out), flush=True)
    del item, ps

if 'small' in what:
    n_items, n = 10_000_001, 64
    item = table(n_items, 8)
    counts = zipf_counts(n_items, 100_000_000)
    ps = ra.PopularSamplerModel(counts).to(dev)
    out = {}
    for B in (4096, 16384, 65536):
        uid = torch.randint(1, user.shape[0], (B,), device=dev, generator=g)
        pos = torch.randint(1, n_items, (B,), device=dev, generator=g)
        buf = {}
        def f():
            buf['o'] = ra.ops.fused_forward(item, user, n, out=buf.get('o'), query_index=uid, pos_ids=pos, fused_bpr=True,
                                            sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs())
        def f_nomean():
            buf['p'] = ra.ops.fused_forward(item, user, n, out=buf.get('p'), query_index=uid, pos_ids=pos, fused_bpr=True,
                                            want_mean=False, sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs())
        alg = bytes_per_triplet(d, n, True, fused_loss=True) * B * n
        us, us2 = timeit(f, 100), timeit(f_nomean, 100)
        out[f'B={B}'] = {'event_us': round(us, 1), 'frac': round(alg / us / 1e3 / 8000, 4), 'nomean_us': round(us2, 1)}
    res['small_batch'] = out
    print('small', json.dumps(out), flush=True)
print('RESULT', json.dumps(res))
