import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
from bench import zipf_counts
dev = torch.device('cuda', 0)
N, U, d, n = 10_000_001, 1_000_001, 128, 64
counts = zipf_counts(N, 100_000_000)
m = ra.BaseRetriever({'model': {'embed_dim': d}, 'train': {'negative_count': n}},
                     item_encoder=torch.nn.Embedding(N, d, padding_idx=0), query_encoder=torch.nn.Embedding(U, d, padding_idx=0),
                     sampler=ra.PopularSamplerModel(counts), loss=ra.BPRLoss())
m.fiid, m.fuid, m.frating = 'item_id', 'user_id', 'rating'
m.item_fields, m.query_fields, m.neg_count = {'item_id'}, {'user_id'}, n
m._init_parameter(); m.to(dev)
for B in (4096, 65536):
    batch = {'user_id': torch.randint(1, U, (B,), device=dev), 'item_id': torch.randint(1, N, (B,), device=dev), 'rating': torch.ones(B, device=dev)}
    for _ in range(10): m.training_step(batch)
    torch.cuda.synchronize()
    ts = []
    for _ in range(100):
        t0 = time.perf_counter(); m.training_step(batch); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(f'B={B}: training_step sync latency median {ts[50]:.3f} ms  p90 {ts[90]:.3f}  max {ts[-1]:.3f}')
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): m.training_step(batch)
    torch.cuda.synchronize(); print(f'        pipelined {(time.perf_counter() - t0) * 10:.3f} ms/step')
