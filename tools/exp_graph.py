import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
from recstudio_amd.graph import GraphedBPRStep
dev = torch.device('cuda', 0)
N, U, d, n, B = 10_000_001, 1_000_001, 128, 64, 4096
item = torch.randn(N, d, device=dev) * 0.02
user = torch.randn(U, d, device=dev) * 0.02
uid = torch.randint(1, U, (B,), device=dev); pos = torch.randint(1, N, (B,), device=dev)
sampler = ra.UniformSampler(N)
def T(fn, reps=300, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
g = GraphedBPRStep(item, user, n, B, sampler, mode='grads')
print('graph.step (2 copies + replay of fwd,bwd,advance): %.1f us' % T(lambda: g.step(uid, pos)))
print('graph.replay only:                                 %.1f us' % T(lambda: g.graph.replay()))
bufs = {}
def eager():
    bufs['o'] = ra.ops.fused_forward(item, user, n, out=bufs.get('o'), fused_bpr=True, want_logp=False, want_query_grad=True,
                                     query_index=uid, pos_ids=pos, sampler=ra._native.SAMPLER_UNIFORM)
    o = bufs['o']
    ra.ops.fused_backward(item, user, o['neg_ids'], o['dneg'], query_index=uid, pos_ids=pos, dpos=o['dpos'],
                          dense_item_grad=False, row_item_grad=True, want_query_grad=False)
print('eager fwd+bwd:                                     %.1f us' % T(eager))
def fwd_only():
    bufs['o'] = ra.ops.fused_forward(item, user, n, out=bufs.get('o'), fused_bpr=True, want_logp=False, want_query_grad=True,
                                     query_index=uid, pos_ids=pos, sampler=ra._native.SAMPLER_UNIFORM)
print('eager fwd only:                                    %.1f us' % T(fwd_only))
# a graph with a single trivial kernel: the floor of a replay
x = torch.zeros(16, device=dev)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    x.add_(1)
torch.cuda.current_stream().wait_stream(s)
gg = torch.cuda.CUDAGraph()
with torch.cuda.graph(gg):
    x.add_(1)
print('replay of a 1-kernel graph:                        %.1f us' % T(lambda: gg.replay()))
print('eager x.add_(1):                                   %.1f us' % T(lambda: x.add_(1)))
