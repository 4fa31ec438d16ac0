"""Full-softmax training step at configs[4] (B = 2048, N = 1e6, d = 128): the parts of the backward."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
from recstudio_amd.scorer import full_lse
dev = torch.device('cuda', 0)
torch.manual_seed(0)
N, d, B = 1_000_001, 128, 2048
w = (torch.randn(N, d, device=dev) * 0.02).requires_grad_(True)
q = (torch.randn(B, d, device=dev) * 0.02).requires_grad_(True)
def timeit(fn, reps=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2]
lse = ra.ops.fullscore(w.detach(), q.detach(), want_lse=True)[1]
scale = torch.full((B,), 1.0 / B, device=dev)
flops = 2.0 * B * d * (N - 1)
t = timeit(lambda: ra.ops.fullscore(w.detach(), q.detach(), want_lse=True))
print(f'forward lse             {t:7.3f} ms  {flops / t / 1e9:6.1f} TFLOP/s')
t = timeit(lambda: ra.ops.fullscore_softmax(w.detach(), q.detach(), lse, scale))
print(f'recompute + write       {t:7.3f} ms  {flops / t / 1e9:6.1f} TFLOP/s')
t = timeit(lambda: ra.ops.fullscore_softmax(w.detach(), q.detach(), lse, scale, want_query_grad=True))
print(f'recompute + write + dQ  {t:7.3f} ms  {2 * flops / t / 1e9:6.1f} TFLOP/s (two products)')
probs = ra.ops.fullscore_softmax(w.detach(), q.detach(), lse, scale)
t = timeit(lambda: probs @ w.detach()[1:])
print(f'dQ by rocBLAS           {t:7.3f} ms  {flops / t / 1e9:6.1f} TFLOP/s')
t = timeit(lambda: probs.t() @ q.detach())
print(f'dX by rocBLAS           {t:7.3f} ms  {flops / t / 1e9:6.1f} TFLOP/s')
del probs
def step():
    w.grad = q.grad = None
    full_lse(q, w).mean().backward()
t = timeit(step, 5)
print(f'training step           {t:7.3f} ms')
