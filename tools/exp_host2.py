"""Host time of the four launches of the sharded forward step at world size 1, stage by stage (perf_counter around each
backend call, no synchronisation inside the loop).  usage: python tools/exp_host2.py [tiny|real]"""
import os, sys, time, json, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
import recstudio_amd as ra
from recstudio_amd import shard
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29579')
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
real = len(sys.argv) > 1 and sys.argv[1] == 'real'
N, U, d, B, n = (12_500_001, 1_000_001, 128, 4096, 1024) if real else (100_001, 10_001, 128, 64, 64)
g = torch.Generator(device=dev).manual_seed(1)
item = torch.empty(N, d, device=dev).normal_(0, 0.02, generator=g)
user = torch.empty(U, d, device=dev).normal_(0, 0.02, generator=g)
uid = torch.randint(1, U, (B,), device=dev, generator=g)
pos = torch.randint(1, N, (B,), device=dev, generator=g)
tbl = shard.ShardedItemTable(item, shard.RowShardPlan(N, 1), 0, dist, check_every=0)
smp = ra.UniformSampler(N)
acc = collections.defaultdict(float)
be = tbl.backend
for name in ('gather_rows', 'sample_route', 'score_segments', 'home'):
    def wrap(fn, name=name):
        def inner(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            acc[name] += time.perf_counter() - t0
            return r
        return inner
    setattr(be, name, wrap(getattr(be, name)))
def step():
    return tbl.sample_and_score(user, uid, pos, n, smp, fused_loss='bpr', want_ids=False, want_grad=True)
for _ in range(20): step()
torch.cuda.synchronize()
acc.clear()
K = 300
t0 = time.perf_counter()
for _ in range(K): step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(json.dumps({'shape': 'real' if real else 'tiny', 'issue_us_per_step': round(t_issue / K * 1e6, 1), 'wall_us_per_step': round(t_all / K * 1e6, 1),
                  'stage_host_us': {k: round(v / K * 1e6, 1) for k, v in acc.items()}}))
dist.destroy_process_group()
