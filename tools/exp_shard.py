"""GPU box, 1 GPU: where does the time of the sharded step go? (world_size-1 RCCL, plan with G virtual shards
is not possible on one GPU, so this measures protocol overhead, not xGMI bandwidth)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
import recstudio_amd as ra
from recstudio_amd import _native as nat
from recstudio_amd.shard import RowShardPlan, ShardedItemTable
from bench import make_workload, zipf_counts
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
N, U, d, n = 10_000_001, 1_000_001, 128, 64
B = int(os.environ.get('B', 65536))
item, user = make_workload(dev, N, U, d)
sampler = ra.PopularSamplerModel(zipf_counts(N, 100_000_000)).to(dev)
g = torch.Generator(device=dev).manual_seed(1)
uid = torch.randint(1, U, (B,), device=dev, generator=g)
pos = torch.randint(1, N, (B,), device=dev, generator=g)
table = ShardedItemTable(item, RowShardPlan(N, 1), 0, dist)
hb = table.backend
def T(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print('whole step           %.3f ms' % T(lambda: table.sample_and_score(user, uid, pos, n, sampler)))
q = hb.gather_rows(user, uid)
lp, neg, ln = hb.sample(sampler, B, n, dev, pos)
print('user gather          %.3f ms' % T(lambda: hb.gather_rows(user, uid)))
print('sample               %.3f ms' % T(lambda: hb.sample(sampler, B, n, dev, pos)))
print('all_gather q         %.3f ms' % T(lambda: table._all_gather_rows(q)))
print('count                %.3f ms' % T(lambda: hb.count(pos, neg, table.plan)))
counts = hb.count(pos, neg, table.plan)
print('exchange counts+sync %.3f ms' % T(lambda: table._exchange_counts(counts)))
sc, rc = table._exchange_counts(counts)
starts = torch.tensor([0] + sc[:-1], dtype=torch.int64).cumsum(0)
print('route                %.3f ms' % T(lambda: hb.route(pos, neg, table.plan, 0, starts)))
keys, positions = hb.route(pos, neg, table.plan, 0, starts)
print('all_to_all keys      %.3f ms' % T(lambda: table._all_to_all(keys, rc, sc)))
print('score keys           %.3f ms' % T(lambda: hb.score_keys(item, q, keys)))
s = hb.score_keys(item, q, keys)
print('all_to_all scores    %.3f ms' % T(lambda: table._all_to_all(s, sc, rc)))
print('scatter              %.3f ms' % T(lambda: hb.scatter(s, positions, B * (n + 1))))
dist.destroy_process_group()
