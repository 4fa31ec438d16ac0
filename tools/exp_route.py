"""Routing kernels alone (rsa_shard_route_fixed / rsa_shard_count) for G = 1, 2, 8 owners at the configs[3] shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
from recstudio_amd import shard
dev = torch.device('cuda', 0)
hb = shard.HipBackend()
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2] * 1e3
N = 100_000_001
for B, n in ((4096, 1024), (65536, 64), (2048, 1024)):
    g = torch.Generator(device=dev).manual_seed(1)
    pos = torch.randint(1, N, (B,), device=dev, generator=g)
    neg = torch.randint(1, N, (B, n), device=dev, generator=g)
    for G in (1, 2, 8):
        plan = shard.RowShardPlan(N, G)
        cap = (int(B * (n + 1) / G * 1.08) + 4096 + 255) // 256 * 256
        flag = hb.new_flag(dev)
        us = timeit(lambda: hb.route_fixed(pos, neg, plan, 0, cap, flag))
        uc = timeit(lambda: hb.count(pos, neg, plan))
        print(f'B={B} n={n} G={G}: route_fixed {us:7.1f} us   count {uc:7.1f} us   overflow={int(flag)}', flush=True)
