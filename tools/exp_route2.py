"""The v2 routing launch alone (rsa_shard_sample_route) at the configs[3] per-GPU shape for 1 / 8 owners and 1 / 8 banks, and
the home kernel alone: kernel time by events.  usage: python tools/exp_route2.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
from recstudio_amd import shard
dev = torch.device('cuda', 0)
hb = shard.HipBackend()


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for B, n in ((4096, 1024), (65536, 64)):
    pos = torch.randint(1, 100_000_001, (B,), device=dev)
    for G in (1, 8):
        plan = shard.RowShardPlan(100_000_001, G)
        for smp_name in ('uniform', 'given'):
            spec = hb.sampler_spec(ra.UniformSampler(100_000_001)) if smp_name == 'uniform' else None
            neg = torch.randint(1, 100_000_001, (B, n), device=dev) if spec is None else None
            for banks in (1, 8):
                st = hb.new_state(dev)
                gen = torch.Generator(device=dev).manual_seed(1)
                counts = hb.sample_route(st, plan, 0, pos, n, 1, 0, spec, gen, neg=neg, count_only=True, banks=banks)
                cap = (int(int(counts.max()) * 1.08) + 4096 // banks + 255) // 256 * 256
                us = timeit(lambda: hb.sample_route(st, plan, 0, pos, n, 1, cap, spec, gen, neg=neg, banks=banks))
                uc = timeit(lambda: hb.sample_route(st, plan, 0, pos, n, 1, 0, spec, gen, neg=neg, count_only=True, banks=banks))
                print(f'B={B} n={n} G={G} {smp_name:8s} banks={banks}: route {us:7.1f} us   count-only {uc:7.1f} us', flush=True)
    r = hb.sample_route(hb.new_state(dev), shard.RowShardPlan(100_000_001, 1), 0, pos, n, 1, B * (n + 1), hb.sampler_spec(ra.UniformSampler(100_000_001)),
                        torch.Generator(device=dev).manual_seed(1), banks=1)
    sc = torch.randn(r['send'].numel(), device=dev)
    for kw in (dict(), dict(loss='bpr'), dict(loss='bpr', want_grad=True), dict(loss='bpr', want_grad=True, want_scores=False),
               dict(loss='bpr', want_dsend=True, want_scores=False)):
        us = timeit(lambda: hb.home(sc, r['slot_of'], B, n, **kw))
        print(f'B={B} n={n} home {kw}: {us:7.1f} us', flush=True)
