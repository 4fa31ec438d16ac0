"""One variant of bpr_sgd_step at the headline shape for rocprofv3.  usage: python tools/exp_sgd_prof.py {0|1} {uniform|popular}"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
from bench import zipf_counts
dev = torch.device('cuda', 0)
mode, name = bool(int(sys.argv[1])), sys.argv[2]
N, U, d, B, n = 10_000_001, 1_000_001, 128, 65536, 64
g = torch.Generator(device=dev).manual_seed(1)
item = torch.empty(N, d, device=dev).normal_(0, 0.02, generator=g)
user = torch.empty(U, d, device=dev).normal_(0, 0.02, generator=g)
uid = torch.randint(1, U, (B,), device=dev, generator=g)
pos = torch.randint(1, N, (B,), device=dev, generator=g)
path = f'/tmp/rsa_ps_1e7.pt'
if name == 'popular':
    if os.path.exists(path):
        smp = torch.load(path, weights_only=False).to(dev)
    else:
        smp = ra.PopularSamplerModel(zipf_counts(N, 100_000_000))
        torch.save(smp, path)
        smp = smp.to(dev)
else:
    smp = ra.UniformSampler(N)
for _ in range(13):
    ra.fused.bpr_sgd_step(item, user, n, 1e-3, user_ids=uid, pos_ids=pos, sampler=smp, in_forward=mode)
torch.cuda.synchronize()
print(json.dumps({'shape': f'sgd_{name}_{int(mode)}', 'steps': 10, 'warmup': 3}))
