"""A/B of the segment-form scoring kernel (rsa_shard_score_segments) between library builds IN ONE PROCESS: the same row
block, the same routed segments, alternating rounds of launches through each library (process-to-process placement of the
6.4 GB block moves the step by +-4 %, more than the variants differ).  usage: python tools/exp_seg.py name=path.so ..."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recstudio_amd as ra                      # noqa: E402
from recstudio_amd import _native as nat        # noqa: E402
from recstudio_amd.shard import HipBackend, RowShardPlan   # noqa: E402
from recstudio_amd._native import ptr           # noqa: E402

dev = torch.device('cuda', 0)
N, d = int(os.environ.get('ITEMS', 12_500_001)), 128
n, B, world = int(os.environ.get('NEG', 1024)), int(os.environ.get('B', 4096)), int(os.environ.get('WORLD', 1))
g = torch.Generator(device=dev).manual_seed(1)
plan = RowShardPlan(N * world, world)
item = torch.empty(plan.n_local(0), d, device=dev).normal_(0, 0.02, generator=g)
q_all = torch.empty(world * B, d, device=dev).normal_(0, 0.02, generator=g)
pos = torch.randint(1, N * world, (B,), device=dev, generator=g)
hb = HipBackend()
st = hb.new_state(dev)
spec = hb.sampler_spec(ra.UniformSampler(N * world))
gen = torch.Generator(device=dev).manual_seed(3)
counts = hb.sample_route(st, plan, 0, pos, n, 1, 0, spec, gen, count_only=True, banks=hb.BANKS)
cap = (int(counts.max() * 1.08) + 4096 // hb.BANKS + 255) // 256 * 256
r = hb.sample_route(st, plan, 0, pos, n, 1, cap, spec, gen, banks=hb.BANKS)
stride, nseg = r['stride'], world * hb.BANKS
send = r['send'].view(nseg, stride)
# as the owner sees it: `world` sources each sending the segments routed to owner 0 (here: source 0's, repeated)
recv = send[:hb.BANKS].repeat(world, 1).contiguous().view(-1) if world > 1 else r['send']
scores = torch.empty(nseg * stride, dtype=torch.float32, device=dev)
libs = {'default': nat.lib()}
for spec_ in sys.argv[1:]:
    name, path = spec_.split('=')
    h = ctypes.CDLL(path)
    fn = h.rsa_shard_score_segments
    fn.restype, fn.argtypes = nat.SIGNATURES['rsa_shard_score_segments']
    libs[name] = h
stream = ra.ops._stream()


def launch(h):
    rc = h.rsa_shard_score_segments(ptr(item), item.shape[0], d, ptr(q_all), q_all.shape[0], ptr(recv), nseg, stride, ptr(scores),
                                    None, None, stream)
    assert rc == 0


res = {k: [] for k in libs}
for h in libs.values():
    for _ in range(20):
        launch(h)
torch.cuda.synchronize()
for rnd in range(6):
    for name, h in libs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            launch(h)
        e1.record()
        torch.cuda.synchronize()
        res[name].append(round(e0.elapsed_time(e1) / 50 * 1e3, 1))
print(json.dumps({'world': world, 'B': B, 'n': n, 'us_per_launch': res}))
