"""Why does the headline kernel take 411 us in one process and 476 us in the next on the same box?
One process, the SAME binary and shapes, the headline launch timed with HIP events over several ALLOCATIONS of its tables: if the
time moves with the allocation (which physical pages the 5 GB table / the 134 MB bucket lines landed on) and not with the
process or the profiler, the spread is placement.  `python tools/exp_placement.py [rounds]` prints one JSON line."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recstudio_amd as ra                      # noqa: E402
from recstudio_amd import _native as nat        # noqa: E402
from bench import zipf_counts                   # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device('cuda', 0)
N, U, d, B, n = 10_000_001, 1_000_001, 128, 65536, 64
gen = torch.Generator(device=dev).manual_seed(100)
uid = torch.randint(1, U, (B,), device=dev, generator=gen)
pos = torch.randint(1, N, (B,), device=dev, generator=gen)
ps_host = ra.PopularSamplerModel(zipf_counts(N, 100_000_000))


def measure(item, user, ps, k=60):
    kw = dict(query_index=uid, pos_ids=pos, sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs())
    buf = {}
    for _ in range(150):
        buf['o'] = ra.ops.fused_forward(item, user, n, out=buf.get('o'), fused_bpr=True, want_mean=False, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        buf['o'] = ra.ops.fused_forward(item, user, n, out=buf['o'], fused_bpr=True, want_mean=False, **kw)
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / k * 1e3, 1)


out = {'pid': os.getpid(), 'profiled': bool(os.environ.get('ROCPROFILER_REGISTER_FORCE_LOAD') or os.environ.get('ROCP_TOOL_LIBRARIES')), 'us': []}
import copy                                     # noqa: E402
pads = []
for r in range(rounds):
    g = torch.Generator(device=dev).manual_seed(1)
    item = torch.empty(N, d, device=dev).normal_(0, 0.02, generator=g)
    user = torch.empty(U, d, device=dev).normal_(0, 0.02, generator=g)
    ps = copy.deepcopy(ps_host).to(dev)
    a = measure(item, user, ps)
    b = measure(item, user, ps)                 # the same allocation again: run-to-run noise inside one placement
    out['us'].append({'alloc': r, 'first': a, 'again': b, 'item_ptr_mod_2MiB': item.data_ptr() % (2 << 20),
                      'lines_ptr_mod_2MiB': ps.cdf_lines.data_ptr() % (2 << 20) if getattr(ps, 'cdf_lines', None) is not None else None})
    del item, user, ps
    # shift where the next allocation lands: keep an odd-sized block alive
    pads.append(torch.empty((r + 1) * 37_000_003, dtype=torch.uint8, device=dev))
    torch.cuda.empty_cache()
print(json.dumps(out))

# ---- second question: does the SAME allocation get faster after the GPU has been kept busy for a while (clock / power state)?
if os.environ.get('SUSTAIN'):
    from bench import box_state                 # noqa: E402
    g = torch.Generator(device=dev).manual_seed(1)
    item = torch.empty(N, d, device=dev).normal_(0, 0.02, generator=g)
    user = torch.empty(U, d, device=dev).normal_(0, 0.02, generator=g)
    ps = copy.deepcopy(ps_host).to(dev)
    kw = dict(query_index=uid, pos_ids=pos, sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs())
    bb = {}

    def busy():
        bb['o'] = ra.ops.fused_forward(item, user, n, out=bb.get('o'), fused_bpr=True, want_mean=False, **kw)
    res = {'cold_us': measure(item, user, ps), 'cold_box': box_state(busy)}
    q = user[1:2049].contiguous()
    it6 = item[:1_000_001]
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < float(os.environ['SUSTAIN']):       # seconds of full-catalog GEMMs (MFMA + HBM)
        for _ in range(10):
            ra.ops.fullscore(it6, q, want_lse=True)
        torch.cuda.synchronize()
    res['after_gemm_us'] = measure(item, user, ps)
    res['after_gemm_box'] = box_state(busy)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < float(os.environ['SUSTAIN']):       # seconds of the headline launch itself
        for _ in range(100):
            busy()
        torch.cuda.synchronize()
    res['after_self_us'] = measure(item, user, ps)
    # fresh (user, positive) rows every launch, as bench.py's timed region does
    uid_all = torch.randint(1, U, (64, B), device=dev, generator=gen)
    pos_all = torch.randint(1, N, (64, B), device=dev, generator=gen)
    buf = {}
    for i in range(64):
        buf['o'] = ra.ops.fused_forward(item, user, n, out=buf.get('o'), fused_bpr=True, want_mean=False, **dict(kw, query_index=uid_all[i], pos_ids=pos_all[i]))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(64):
        buf['o'] = ra.ops.fused_forward(item, user, n, out=buf['o'], fused_bpr=True, want_mean=False, **dict(kw, query_index=uid_all[i], pos_ids=pos_all[i]))
    e1.record()
    torch.cuda.synchronize()
    res['fresh_batches_us'] = round(e0.elapsed_time(e1) / 64 * 1e3, 1)
    print(json.dumps(res))
