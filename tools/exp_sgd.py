"""bpr_sgd_step at the headline shape: all-sorted vs in-forward update of the solo rows.  usage: python tools/exp_sgd.py"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
from bench import zipf_counts, time_gpu
dev = torch.device('cuda', 0)
N, U, d, B, n = int(os.environ.get('ITEMS', 10_000_001)), 1_000_001, 128, 65536, 64
g = torch.Generator(device=dev).manual_seed(1)
item = torch.empty(N, d, device=dev).normal_(0, 0.02, generator=g)
user = torch.empty(U, d, device=dev).normal_(0, 0.02, generator=g)
uid = torch.randint(1, U, (B,), device=dev, generator=g)
pos = torch.randint(1, N, (B,), device=dev, generator=g)
for name, smp in (('uniform', ra.UniformSampler(N)), ('popular', ra.PopularSamplerModel(zipf_counts(N, 100_000_000)).to(dev))):
    for mode in (False, True):
        t = time_gpu(lambda: ra.fused.bpr_sgd_step(item, user, n, 1e-3, user_ids=uid, pos_ids=pos, sampler=smp, in_forward=mode), 30, 5) * 1e3
        _, ids = ra.fused.bpr_sgd_step(item, user, n, 1e-3, user_ids=uid, pos_ids=pos, sampler=smp, in_forward=mode)
        cnt = torch.bincount(torch.cat([pos, ids.reshape(-1)]), minlength=N)
        solo_elems = int((cnt[ids.reshape(-1)] == 1).sum())
        print(json.dumps({'sampler': name, 'in_forward': mode, 'sgd_step_ms': round(t, 4), 'solo_share_of_elements': round(solo_elems / ids.numel(), 3)}), flush=True)
