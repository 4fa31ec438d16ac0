"""B = 4096 launches only (for a rocprofv3 kernel trace: duration of the kernel itself vs the event-to-event time)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
from recstudio_amd import _native as nat
from bench import zipf_counts
dev = torch.device('cuda', 0)
d, n = 128, 64
B = int(os.environ.get('B', 4096))
N = 10_000_001
g = torch.Generator(device=dev).manual_seed(1)
item = torch.empty(N, d, device=dev).normal_(0, 0.02, generator=g)
user = torch.empty(1_000_001, d, device=dev).normal_(0, 0.02, generator=g)
uid = torch.randint(1, 1_000_001, (B,), device=dev, generator=g)
pos = torch.randint(1, N, (B,), device=dev, generator=g)
ps = (torch.load("/tmp/rsa_ps_1e7.pt", weights_only=False) if os.path.exists("/tmp/rsa_ps_1e7.pt") else ra.PopularSamplerModel(zipf_counts(N, 100_000_000))).to(dev)
res = {}
for name, kw in (('popular', dict(sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs())), ('uniform', dict(sampler=nat.SAMPLER_UNIFORM)),
                 ('given', dict(neg_ids=torch.randint(1, N, (B, n), device=dev, generator=g)))):
    buf = {}
    def f():
        buf['o'] = ra.ops.fused_forward(item, user, n, out=buf.get('o'), query_index=uid, pos_ids=pos, fused_bpr=True, want_mean=False, **kw)
    for _ in range(20): f()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
    for a, b in evs:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    res[name] = round(t[50] * 1e3, 1)
print(json.dumps(res))
