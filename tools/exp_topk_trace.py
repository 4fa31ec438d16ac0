import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
dev = torch.device('cuda', 0)
torch.manual_seed(0)
item = torch.randn(1_000_001, 128, device=dev) * 0.02
q = torch.randn(2048, 128, device=dev) * 0.02
for _ in range(4):
    ra.ops.fullscore(item, q, want_lse=True, k=100)
torch.cuda.synchronize()
