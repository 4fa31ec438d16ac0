"""Why does bench.py time the queue figure (16 x B=4096 in one launch) at 406 us and the sharded world-1 forward at 399 us when
tools/prof_shapes.py -- with or without rocprofv3 -- times the same shapes at 454 and 446 us on the same box, while the headline
agrees (466 / 469)?  One process, HIP events, 100 launches per figure, the candidates side by side and repeated:
  headline through ops.fused_forward / through ops.FusedStep / the queue with two id sets / the sharded step on a fresh block
  and on a block carved out of a freed 51 GB allocation (what bench.py's allocator state looks like by then).
`python tools/exp_outliers.py` prints one JSON line."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recstudio_amd as ra                      # noqa: E402
from recstudio_amd import _native as nat        # noqa: E402
from bench import zipf_counts, prewarm          # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
N, U, d, B, n = 10_000_001, 1_000_001, 128, 65536, 64


def table(rows, seed):
    t = torch.empty(rows, d, device=dev).normal_(0, 0.02, generator=torch.Generator(device=dev).manual_seed(seed))
    t[0] = 0
    return t


def timed(fn, k=100):
    prewarm(fn, 1.5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / k * 1e3, 1)


item, user = table(N, 1), table(U, 2)
ps = ra.PopularSamplerModel(zipf_counts(N, 100_000_000)).to(dev)
gen = torch.Generator(device=dev).manual_seed(100)
uid = torch.randint(1, U, (B,), device=dev, generator=gen)
pos = torch.randint(1, N, (B,), device=dev, generator=gen)
kw = dict(sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs())
buf = {}


def headline():
    buf['o'] = ra.ops.fused_forward(item, user, n, out=buf.get('o'), fused_bpr=True, query_index=uid, pos_ids=pos, **kw)


frozen = ra.ops.FusedStep(item, user, n, fused_bpr=True, query_index=uid, pos_ids=pos, **kw)
queue_same = ra.ops.FusedStep(item, user, n, fused_bpr=True, n_batches=16, query_index=uid, pos_ids=pos, **kw)
g77 = torch.Generator(device=dev).manual_seed(77)
uq = torch.randint(1, U, (B,), device=dev, generator=g77)
pq = torch.randint(1, N, (B,), device=dev, generator=g77)
queue_77 = ra.ops.FusedStep(item, user, n, fused_bpr=True, n_batches=16, query_index=uq, pos_ids=pq, **kw)
out = {'rounds': []}
for r in range(3):
    out['rounds'].append({'headline_fused_forward': timed(headline), 'headline_frozen': timed(frozen),
                          'queue16_same_ids': timed(queue_same), 'queue16_seed77_ids': timed(queue_77)})
del frozen, queue_same, queue_77, buf

import torch.distributed as dist                # noqa: E402
from recstudio_amd import shard                 # noqa: E402
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29579')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
n_blk, n1k, b1k = 12_500_001, 1024, 4096
us = ra.UniformSampler(n_blk)
u1, p1 = uid[:b1k].contiguous(), torch.randint(1, n_blk, (b1k,), device=dev, generator=gen)


def sharded(blk):
    tbl = shard.ShardedItemTable(blk, shard.RowShardPlan(n_blk, 1), 0, dist, check_every=0)

    def st():
        tbl.sample_and_score(user, u1, p1, n1k, us, fused_loss='bpr', want_ids=False, want_grad=True)
    st()
    return [timed(st), timed(st)]


blk = table(n_blk, 9)
out['sharded_fresh_block'] = sharded(blk)
out['fresh_block_ptr'] = hex(blk.data_ptr())
del blk
big = torch.empty(100_000_001, d, device=dev)
big.zero_()
del big                                          # stays in torch's caching allocator: the next block is carved out of it
blk = table(n_blk, 9)
out['sharded_block_from_freed_51GB'] = sharded(blk)
out['carved_block_ptr'] = hex(blk.data_ptr())
print(json.dumps(out))
