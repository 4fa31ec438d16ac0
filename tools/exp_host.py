"""Host-side cost of one sharded step (tiny shapes: the GPU work is negligible) under cProfile."""
import os, sys, cProfile, pstats, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
import recstudio_amd as ra
from recstudio_amd import _native as nat, shard
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29578')
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
N, U, d, B, n = 100_001, 10_001, 128, 64, 64
g = torch.Generator(device=dev).manual_seed(1)
item = torch.empty(N, d, device=dev).normal_(0, 0.02, generator=g)
user = torch.empty(U, d, device=dev).normal_(0, 0.02, generator=g)
uid = torch.randint(1, U, (B,), device=dev, generator=g)
pos = torch.randint(1, N, (B,), device=dev, generator=g)
tbl = shard.ShardedItemTable(item, shard.RowShardPlan(N, 1), 0, dist, chunks=int(os.environ.get('CHUNKS', 1)))
smp = ra.UniformSampler(N)
def step():        # the bench's sharded_world1 step
    return tbl.sample_and_score(user, uid, pos, n, smp, fused_loss='bpr', want_ids=False, want_grad=True)
for _ in range(20): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
dist.destroy_process_group()
