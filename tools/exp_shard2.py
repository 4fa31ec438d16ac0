"""Sharded step at world size 1 (RCCL), configs[3] per-GPU shape: wall time per step (events + host clock) -- run under
rocprofv3 --kernel-trace --stats for the per-kernel view.  Env: NEG, B, CHUNKS, EXCHANGE, SAMPLER (uniform | popular),
STEP (fwd: forward + fused BPR loss, the bench's sharded_world1 step | old: scores out + stand-alone loss kernel |
train: forward + loss + gradient exchange + in-place SGD)."""
import os, sys, json, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
import recstudio_amd as ra
from recstudio_amd import _native as nat, shard
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
N, U, d = int(os.environ.get('ITEMS', 12_500_001)), 1_000_001, 128
n, B = int(os.environ.get('NEG', 1024)), int(os.environ.get('B', 4096))
mode = os.environ.get('STEP', 'fwd')
g = torch.Generator(device=dev).manual_seed(1)
item = torch.empty(N, d, device=dev).normal_(0, 0.02, generator=g)
user = torch.empty(U, d, device=dev).normal_(0, 0.02, generator=g)
uid = torch.randint(1, U, (B,), device=dev, generator=g)
pos = torch.randint(1, N, (B,), device=dev, generator=g)
tbl = shard.ShardedItemTable(item, shard.RowShardPlan(N, 1), 0, dist, exchange=os.environ.get('EXCHANGE', 'fixed'),
                             chunks=int(os.environ.get('CHUNKS', 1)), check_every=int(os.environ.get('CHECK_EVERY', 16)))
if os.environ.get('SAMPLER', 'uniform') == 'popular':
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import zipf_counts
    smp = ra.PopularSamplerModel(zipf_counts(N, 100_000_000)).to(dev)
else:
    smp = ra.UniformSampler(N)
if mode == 'train':
    tower = torch.nn.Embedding(U, d).to(dev)
    trainer = shard.ShardedRetriever(tbl, tower, smp, ra.BPRLoss(), n, item_sgd_lr=0.05, query_sgd_lr=0.05)
    step = lambda: trainer.training_step(uid, pos)
elif mode == 'old' or tbl.exchange == 'exact':
    def step():
        o = tbl.sample_and_score(user, uid, pos, n, smp)
        return ra.ops.pairwise_loss(nat.LOSS_BPR, o['pos_score'], o['neg_score'], want_grad=True)
else:
    step = lambda: tbl.sample_and_score(user, uid, pos, n, smp, fused_loss='bpr', want_ids=False, want_grad=True)
for _ in range(5): step()
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
for a, b in evs:
    a.record(); step(); b.record()
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) for a, b in evs)
t0 = time.perf_counter()
for _ in range(50): step()
t_issue = (time.perf_counter() - t0) / 50 * 1e3
torch.cuda.synchronize()
t_host = (time.perf_counter() - t0) / 50 * 1e3
tbl.check_overflow()
print(json.dumps({'B': B, 'n': n, 'mode': mode, 'chunks': tbl.chunks, 'sampler': type(smp).__name__, 'event_median_ms': round(t[25], 4),
                  'host_ms': round(t_host, 4), 'python_issue_ms': round(t_issue, 4)}))
dist.destroy_process_group()
