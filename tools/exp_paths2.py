"""Second sweep: loaders (rows/s delivered on the device), SASRec-shaped training step ([B, L] targets), evaluation
(topk with history + metrics)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
from recstudio_amd import dataset as D
dev = torch.device('cuda', 0)
def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return time.perf_counter() - t0, r
# --- loaders on a synthetic interaction table
rng = np.random.default_rng(0)
n_users, n_items, n_inter = 200_000, 1_000_000, 20_000_000
def make(cls, max_seq_len=None):
    ds = cls.__new__(cls)
    ds.config = {'max_seq_len': max_seq_len, 'low_rating_thres': None}
    ds.fuid, ds.fiid, ds.frating, ds.ftime = 'user_id', 'item_id', 'rating', 'timestamp'
    ds.use_field = {'user_id', 'item_id', 'rating'}
    ds.eval_mode = False
    return ds
try:
    ds = make(D.TripletDataset)
    u = np.sort(rng.integers(1, n_users, n_inter))
    ds.inter_feat = {'user_id': torch.from_numpy(u), 'item_id': torch.from_numpy(rng.integers(1, n_items, n_inter)),
                     'rating': torch.ones(n_inter)}
    ds.data_index = torch.arange(n_inter)
    t, ld = wall(lambda: ds.device_train_loader(65536, shuffle=True, device=dev))
    def epoch():
        c = 0
        for b in ld: c += b['item_id'].numel()
        return c
    epoch()
    t2, c = wall(epoch)
    print(f'TripletDataset device loader: {c} rows in {t2 * 1e3:.1f} ms = {c / t2 / 1e6:.1f} M rows/s ({t2 / len(ld) * 1e3:.3f} ms per batch of 65536; setup {t:.2f} s)', flush=True)
except Exception as e:
    print('triplet loader: ERROR', repr(e)[:300])
try:
    ds = make(D.SeqDataset, 50)
    n_inter2 = 4_000_000
    ds.inter_feat = {'user_id': torch.from_numpy(np.sort(rng.integers(1, 50_000, n_inter2))),
                     'item_id': torch.from_numpy(rng.integers(1, n_items, n_inter2)), 'rating': torch.ones(n_inter2)}
    uu = ds.inter_feat['user_id']
    starts = torch.cat([torch.zeros(1, dtype=torch.int64), (uu[1:] != uu[:-1]).nonzero().view(-1) + 1])
    user_start = torch.zeros(n_inter2, dtype=torch.int64); user_start[starts] = starts
    user_start = torch.cummax(user_start, 0).values
    end = torch.arange(n_inter2)
    keep = end > user_start
    ds.data_index = torch.stack([uu[keep], torch.maximum(user_start[keep], end[keep] - 50), end[keep]], 1)
    ld = ds.device_train_loader(8192, shuffle=True, device=dev)
    def epoch2():
        c = 0
        for b in ld: c += b['item_id'].numel()
        return c
    epoch2()
    t2, c = wall(epoch2)
    print(f'SeqDataset device loader (L <= 50): {c} prefixes in {t2 * 1e3:.1f} ms = {c / t2 / 1e6:.2f} M prefixes/s ({t2 / len(ld) * 1e3:.3f} ms per batch of 8192)', flush=True)
except Exception as e:
    import traceback; traceback.print_exc()
    print('seq loader: ERROR', repr(e)[:300])
# --- evaluation: topk with history + metric step
try:
    N, U, d = 1_000_001, 100_001, 128
    m = ra.BaseRetriever(None, item_encoder=torch.nn.Embedding(N, d, padding_idx=0), query_encoder=torch.nn.Embedding(U, d, padding_idx=0),
                         scorer=ra.InnerProductScorer())
    m.fiid, m.fuid, m.frating = 'item_id', 'user_id', 'rating'
    m.item_fields, m.query_fields = {'item_id'}, {'user_id'}
    m.to(dev); m._update_item_vector()
    for B, H in ((2048, 200), (2048, 1500)):
        uid = torch.randint(1, U, (B,), device=dev)
        hist = torch.randint(1, N, (B, H), device=dev).sort(-1).values
        f = lambda: m.topk({'user_id': uid}, 100, hist)
        with torch.no_grad():
            f(); t, _ = wall(lambda: [f() for _ in range(5)])
        print(f'topk B={B} k=100 hist={H}: {t / 5 * 1e3:.3f} ms', flush=True)
except Exception as e:
    import traceback; traceback.print_exc()
