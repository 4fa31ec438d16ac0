"""GPU box: end-to-end plugin-level timings (BaseRetriever.training_step + backward + optimizer) to spot host or
framework overheads around the kernels."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
from bench import zipf_counts
dev = torch.device('cuda', 0)
def T(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for N, U in ((1_000_001, 100_001), (10_000_001, 1_000_001)):
    d, n = 128, 64
    counts = zipf_counts(N, 100_000_000 if N > 2_000_000 else 10_000_000)
    for sparse in (False, True):
        m = ra.BaseRetriever({'model': {'embed_dim': d}, 'train': {'negative_count': n, 'sparse_grad': sparse}},
                             item_encoder=torch.nn.Embedding(N, d, padding_idx=0, sparse=sparse),
                             query_encoder=torch.nn.Embedding(U, d, padding_idx=0, sparse=sparse),
                             sampler=ra.PopularSamplerModel(counts), loss=ra.BPRLoss())
        m.fiid, m.fuid, m.frating = 'item_id', 'user_id', 'rating'
        m.item_fields, m.query_fields, m.neg_count = {'item_id'}, {'user_id'}, n
        m._init_parameter()
        m.to(dev)
        opt = (torch.optim.SparseAdam(m.parameters(), lr=1e-3) if sparse else torch.optim.SGD(m.parameters(), lr=1e-3))
        for B in (4096, 65536):
            batch = {'user_id': torch.randint(1, U, (B,), device=dev), 'item_id': torch.randint(1, N, (B,), device=dev),
                     'rating': torch.ones(B, device=dev)}
            def fwd():
                return m.training_step(batch)
            def fwd_bwd():
                opt.zero_grad(set_to_none=True)
                m.training_step(batch).backward()
            def full():
                opt.zero_grad(set_to_none=True)
                m.training_step(batch).backward()
                opt.step()
            print(f'N={N} sparse={sparse} B={B}: training_step {T(fwd):.3f} ms   +backward {T(fwd_bwd):.3f} ms   +optimizer({type(opt).__name__}) {T(full, 5, 2):.3f} ms', flush=True)
        del m, opt
        torch.cuda.empty_cache()
# evaluation: topk with history
N, d = 1_000_001, 128
item = torch.nn.Embedding(N, d, padding_idx=0)
user = torch.nn.Embedding(100_001, d, padding_idx=0)
m = ra.BaseRetriever(None, item_encoder=item, query_encoder=user, scorer=ra.InnerProductScorer())
m.fiid, m.fuid, m.frating = 'item_id', 'user_id', 'rating'
m.item_fields, m.query_fields = {'item_id'}, {'user_id'}
m.to(dev)
m._update_item_vector()
for B in (512, 2048):
    uid = torch.randint(1, 100_001, (B,), device=dev)
    hist = torch.randint(1, N, (B, 200), device=dev).sort(-1).values
    with torch.no_grad():
        print(f'topk B={B} k=100 hist=200: {T(lambda: m.topk({"user_id": uid, "user_hist": hist}, 100, hist), 5, 2):.3f} ms', flush=True)
