"""GPU box: a few complete training steps (two-launch step, in-place SGD, lazy Adam) for a rocprofv3 kernel trace."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
from recstudio_amd.fused import FusedBPRAdam, bpr_sgd_step, fused_bpr_loss
from bench import make_workload, zipf_counts
dev = torch.device('cuda', 0)
N, U, d, n, B = 10_000_001, 1_000_001, 128, 64, 65536
item, user = make_workload(dev, N, U, d)
sampler = ra.PopularSamplerModel(zipf_counts(N, 100_000_000)).to(dev)
uid = torch.randint(1, U, (B,), device=dev)
pos = torch.randint(1, N, (B,), device=dev)
fa = FusedBPRAdam(item, user, lr=1e-4)
for _ in range(12):
    bpr_sgd_step(item, user, n, 1e-4, user_ids=uid, pos_ids=pos, sampler=sampler)
    fa.step(n, user_ids=uid, pos_ids=pos, sampler=sampler)
iw, uw = item.requires_grad_(True), user.requires_grad_(True)
for _ in range(6):
    iw.grad = uw.grad = None
    loss, _ = fused_bpr_loss(iw, uw, n, query_index=uid, pos_ids=pos, sampler=sampler)      # dense autograd gradients
    loss.backward()
torch.cuda.synchronize()
