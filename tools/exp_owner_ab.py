"""In-process A/B of the owner-side BPR pass (rsa_shard_owner_bpr_forward: sorts + classification + the walk) between
library builds (tools/build_variant.sh with VARIANT_FILES=rsa_owner): one world-1 sharded training step through the default
library yields the argument block; every variant's entry point is then launched on the SAME buffers, alternating rounds.
usage: python tools/exp_owner_ab.py name=path.so [name=path.so ...]     (the item scale is 0: the table does not drift)"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recstudio_amd as ra                      # noqa: E402
from recstudio_amd import _native as nat        # noqa: E402
from recstudio_amd import shard                 # noqa: E402
import torch.distributed as dist                # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29579')
dist.init_process_group('gloo', rank=0, world_size=1)
n_blk, n_neg, B, d, U = 12_500_001, 1024, 4096, 128, 1_000_001
item = torch.empty(n_blk, d, device=dev).normal_(0, 0.02)
item[0] = 0
tbl = shard.ShardedItemTable(item, shard.RowShardPlan(n_blk, 1), 0, dist, check_every=0)
tower = torch.nn.Embedding(U, d).to(dev)
uid = torch.randint(1, U, (B,), device=dev)
pos = torch.randint(1, n_blk, (B,), device=dev)
grabbed = {}
fwd = tbl.backend.owner_bpr_forward
tbl.backend.owner_bpr_forward = lambda *a, **k: grabbed.setdefault('ctx', fwd(*a, **k))
trainer = shard.ShardedRetriever(tbl, tower, ra.UniformSampler(n_blk), ra.BPRLoss(), n_neg, item_sgd_lr=0.0, query_sgd_lr=0.0)
trainer.training_step(uid, pos)
torch.cuda.synchronize()
args = grabbed['ctx']['args']
libs = {'default': nat.LIB_PATH}
for a in sys.argv[1:]:
    k, v = a.split('=', 1)
    libs[k] = v
fns = {}
for k, path in libs.items():
    h = ctypes.CDLL(path)
    fn = h.rsa_shard_owner_bpr_forward
    fn.restype, fn.argtypes = nat.SIGNATURES['rsa_shard_owner_bpr_forward']
    fns[k] = fn
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
res = {k: [] for k in fns}
for rnd in range(4):
    for k, fn in fns.items():
        for _ in range(5):
            assert fn(ctypes.byref(args), stream) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn(ctypes.byref(args), stream)
        e1.record()
        torch.cuda.synchronize()
        res[k].append(round(e0.elapsed_time(e1) / 30 * 1e3, 1))
print(json.dumps(res))
