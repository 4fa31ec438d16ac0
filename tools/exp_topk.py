import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
dev = torch.device('cuda', 0)
torch.manual_seed(0)
N, d = 1_000_001, 128
item = torch.randn(N, d, device=dev)
def T(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for B in (512, 1024, 2048):
    q = torch.randn(B, d, device=dev)
    hist = torch.randint(1, N, (B, 200), device=dev).sort(-1).values
    for k in (100, 200, 300, 500):
        t = T(lambda: ra.ops.fullscore(item, q, k=k))
        _, _, v, i = ra.ops.fullscore(item, q, k=k)
        t2 = T(lambda: ra.ops.topk_mask_history(v, i, hist, min(100, k)))
        print(f'B={B} k={k}: fullscore {t:.3f} ms   mask_history {t2:.3f} ms', flush=True)
