"""Sharded step at world size 1 (RCCL), configs[3] per-GPU shape: per-stage wall times (events) -- run under
rocprofv3 --kernel-trace --stats for the per-kernel view."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
import recstudio_amd as ra
from recstudio_amd import _native as nat, shard
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
N, U, d = 12_500_001, 1_000_001, 128
n, B = int(os.environ.get('NEG', 1024)), int(os.environ.get('B', 4096))
g = torch.Generator(device=dev).manual_seed(1)
item = torch.empty(N, d, device=dev).normal_(0, 0.02, generator=g)
user = torch.empty(U, d, device=dev).normal_(0, 0.02, generator=g)
uid = torch.randint(1, U, (B,), device=dev, generator=g)
pos = torch.randint(1, N, (B,), device=dev, generator=g)
tbl = shard.ShardedItemTable(item, shard.RowShardPlan(N, 1), 0, dist, exchange=os.environ.get('EXCHANGE', 'fixed'),
                             chunks=int(os.environ.get('CHUNKS', 1)))
smp = ra.UniformSampler(N)
def step():
    o = tbl.sample_and_score(user, uid, pos, n, smp)
    return ra.ops.pairwise_loss(nat.LOSS_BPR, o['pos_score'], o['neg_score'], want_grad=True)
for _ in range(5): step()
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
for a, b in evs:
    a.record(); step(); b.record()
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) for a, b in evs)
import time
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
print(json.dumps({'B': B, 'n': n, 'event_median_ms': round(t[25], 4), 'host_ms': round((time.perf_counter() - t0) / 50 * 1e3, 4)}))
dist.destroy_process_group()
