import torch, sys
sys.path.insert(0, '.')
import recstudio_amd as ra
from recstudio_amd import _native as nat
DEV = 'cuda'
N, U, d, B, n, lr = 200_003, 211, 128, 700, 64, 0.3
g = torch.Generator().manual_seed(1)
iw = torch.randn(N, d, generator=g) * 0.3; iw[0] = 0
uw = torch.randn(U, d, generator=g) * 0.3
uid = torch.randint(1, U, (B,), generator=g).to(DEV); pos = torch.randint(1, N, (B,), generator=g).to(DEV)
smp = ra.UniformSampler(N)
res = []
for mode in (False, True):
    item, user = iw.to(DEV).clone(), uw.to(DEV).clone()
    torch.manual_seed(99)
    loss, ids = ra.fused.bpr_sgd_step(item, user, n, lr, user_ids=uid, pos_ids=pos, in_forward=mode, sampler=smp)
    res.append((loss.clone(), ids.clone(), item, user))
print('loss', res[0][0].item(), res[1][0].item(), (res[0][0] - res[1][0]).item(), 'ids equal', torch.equal(res[0][1], res[1][1]))
torch.manual_seed(99)
a = ra.ops.fused_forward(iw.to(DEV), uw.to(DEV), n, query_index=uid, pos_ids=pos, sampler=nat.SAMPLER_UNIFORM, want_logp=False, fused_bpr=True, want_query_grad=True)
b = ra.ops.fused_forward(iw.to(DEV), uw.to(DEV), n, query_index=uid, pos_ids=pos, neg_ids=a['neg_ids'].clone(), fused_bpr=True, want_query_grad=True)
for k in ('pos_score', 'neg_score', 'row_loss', 'dneg', 'dpos', 'query_grad', 'loss'):
    print(k, (a[k] - b[k]).abs().max().item(), torch.equal(a[k], b[k]))
print('item diff', (res[0][2] - res[1][2]).abs().max().item(), 'user equal', torch.equal(res[0][3], res[1][3]))
