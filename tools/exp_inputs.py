"""tools/prof_shapes.py times the headline launch at ~470 us on every box, on each of four output allocations, while bench.py
and the exp_* scripts get 405-427 us for the same launch.  prof_shapes allocates the user table BEFORE the item table.  Same
order here; then ONE input at a time is replaced by a copy in a fresh allocation: which allocation carries the slow class?
`python tools/exp_inputs.py` prints one JSON line."""
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
order = sys.argv[1] if len(sys.argv) > 1 else 'user_first'


def table(rows, seed):
    t = torch.empty(rows, d, device=dev).normal_(0, 0.02, generator=torch.Generator(device=dev).manual_seed(seed))
    t[0] = 0
    return t


def timed(fn, k=60, warm=0.6):
    prewarm(fn, warm)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / k * 1e3, 1)


if order == 'user_first':
    user = table(U, 3)
    item = table(N, 1)
else:
    item = table(N, 1)
    user = table(U, 3)
ps_host = ra.PopularSamplerModel(zipf_counts(N, 100_000_000))
ps = ps_host.to(dev)
gen = torch.Generator(device=dev).manual_seed(100)
uid = torch.randint(1, U, (B,), device=dev, generator=gen)
pos = torch.randint(1, N, (B,), device=dev, generator=gen)


def run(item, user, ps, uid, pos, out=None):
    kw = dict(query_index=uid, pos_ids=pos, sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs())
    b = {'o': out}

    def step():
        b['o'] = ra.ops.fused_forward(item, user, n, out=b['o'], fused_bpr=True, **kw)
    return timed(step), b['o']


res = {'order': order}
res['as_allocated'], out0 = run(item, user, ps, uid, pos)
res['ptrs'] = {'item': hex(item.data_ptr()), 'user': hex(user.data_ptr()), 'out_neg_ids': hex(out0['neg_ids'].data_ptr())}
item2 = item.clone()
res['item_copy'], _ = run(item2, user, ps, uid, pos, out0)
user2 = user.clone()
res['user_copy'], _ = run(item, user2, ps, uid, pos, out0)
import copy                                     # noqa: E402
ps2 = copy.deepcopy(ps_host).to(dev)
res['sampler_copy'], _ = run(item, user, ps2, uid, pos, out0)
res['ids_copy'], _ = run(item, user, ps, uid.clone(), pos.clone(), out0)
res['all_copies'], out2 = run(item2, user2, ps2, uid.clone(), pos.clone())
res['all_copies_old_outputs'], _ = run(item2, user2, ps2, uid.clone(), pos.clone(), out0)
res['as_allocated_again'], _ = run(item, user, ps, uid, pos, out0)
print(json.dumps(res))
