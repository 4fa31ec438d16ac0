import os, sys, torch
sys.path.insert(0, '/root/repo')
import recstudio_amd as ra
from recstudio_amd import _native as nat
from bench import zipf_counts, time_gpu
dev = torch.device('cuda', 0)
N, U, d, B, n = 10_000_001, 1_000_001, 128, 65536, 64
g = torch.Generator(device=dev).manual_seed(1)
item = torch.empty(N, d, device=dev).normal_(0, 0.02, generator=g)
user = torch.empty(U, d, device=dev).normal_(0, 0.02, generator=g)
uid = torch.randint(1, U, (B,), device=dev, generator=g); pos = torch.randint(1, N, (B,), device=dev, generator=g)
ps = ra.PopularSamplerModel(zipf_counts(N, 100_000_000)).to(dev)
for rep in range(2):
    for name, kw in (('logp', dict(want_logp=True)), ('nologp', dict(want_logp=False))):
        buf = {}
        def f():
            buf['o'] = ra.ops.fused_forward(item, user, n, out=buf.get('o'), query_index=uid, pos_ids=pos, fused_bpr=True, want_mean=True,
                                            sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs(), **kw)
        t = time_gpu(f, 100, 20) * 1e3
        print(name, round(t, 4), flush=True)
