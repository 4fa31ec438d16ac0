"""In-process A/B of the sharded training step (configs[3] per-GPU shape, world size 1) with the query rows' exchange and
update on a second stream (ShardedRetriever(overlap_query_rows=True), the default), behind the apply pass, and one batch
ahead (prepare_step: routing, key exchange, the owner's sorts under the step in front): alternating rounds of whole
training steps, HIP events around each round.  usage: python tools/exp_step_overlap.py"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recstudio_amd as ra                      # noqa: E402
from recstudio_amd import shard                 # noqa: E402
import torch.distributed as dist                # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29581')
dist.init_process_group('gloo', rank=0, world_size=1)
n_blk, n_neg, B, d, U = 12_500_001, 1024, 4096, 128, 1_000_001
item = torch.empty(n_blk, d, device=dev).normal_(0, 0.02)
item[0] = 0
tower = torch.nn.Embedding(U, d).to(dev)
uid = torch.randint(1, U, (B,), device=dev)
pos = torch.randint(1, n_blk, (B,), device=dev)
trainers = {}
for name, flag in (('side_stream', True), ('serial', False)):
    tbl = shard.ShardedItemTable(item, shard.RowShardPlan(n_blk, 1), 0, dist, check_every=0)
    trainers[name] = shard.ShardedRetriever(tbl, tower, ra.UniformSampler(n_blk), ra.BPRLoss(), n_neg, item_sgd_lr=1e-3,
                                            query_sgd_lr=1e-3, overlap_query_rows=flag)
trainers['one_batch_ahead'] = trainers['side_stream']


def run(k, tr, steps):
    if k != 'one_batch_ahead':
        for _ in range(steps):
            tr.training_step(uid, pos)
        return
    ticket = tr.prepare_step(uid, pos)
    for i in range(steps):
        nxt = tr.prepare_step(uid, pos) if i + 1 < steps else None      # second stream: under the step below
        tr.training_step(uid, pos, ticket=ticket)
        ticket = nxt


main_stream = torch.cuda.Stream(priority=-1) if os.environ.get('MAIN_HIGH') == '1' else torch.cuda.current_stream()
torch.cuda.set_stream(main_stream)          # MAIN_HIGH=1: the steps on a high-priority stream, the second stream below it
res = {k: [] for k in trainers}
host = {k: [] for k in trainers}
for rnd in range(5):
    for k, tr in trainers.items():
        run(k, tr, 5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        h0 = time.perf_counter()
        run(k, tr, 30)
        host[k].append((time.perf_counter() - h0) / 30 * 1e3)       # host time to ISSUE a step (no synchronisation inside)
        e1.record()
        torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 30)
print(json.dumps({k: {'min_ms': round(min(v), 4), 'median_ms': round(sorted(v)[len(v) // 2], 4),
                      'host_issue_ms': round(min(host[k]), 4)} for k, v in res.items()}))
