"""Plugin-level sweep: BaseRetriever.training_step + backward for every (loss, scorer, sampler, sampling method) the
path supports, same catalog and batch -- to spot paths far above the byte model (ms per 1 M triplets next to the fused
BPR baseline)."""
import os, sys, time, itertools, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
from bench import zipf_counts
dev = torch.device('cuda', 0)
N, U, d = 1_000_001, 200_001, 128
B = int(os.environ.get('B', 16384))
counts = zipf_counts(N, 10_000_000)
def T(fn, reps=8, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
hist = torch.randint(1, N, (U, 40)).sort(-1).values.to(dev)
def build(loss, scorer, sampler, n, method='none', excluding_hist=False):
    m = ra.BaseRetriever({'model': {'embed_dim': d}, 'train': {'negative_count': n, 'sampling_method': method,
                                                                  'excluding_hist': excluding_hist}},
                         item_encoder=torch.nn.Embedding(N, d, padding_idx=0), query_encoder=torch.nn.Embedding(U, d, padding_idx=0),
                         sampler=sampler, loss=loss, scorer=scorer)
    m.fiid, m.fuid, m.frating = 'item_id', 'user_id', 'rating'
    m.item_fields, m.query_fields, m.neg_count = {'item_id'}, {'user_id'}, n
    m._init_parameter()
    return m.to(dev)
cases = []
for n in (64, 256):
    for lname, loss in (('BPR', ra.BPRLoss), ('SSM', ra.SampledSoftmaxLoss), ('BCE', ra.BinaryCrossEntropyLoss)):
        for sname, scorer in (('ip', ra.InnerProductScorer), ('cos', ra.CosineScorer), ('euc', ra.EuclideanScorer)):
            for pname in ('uniform', 'popular', 'masked'):
                if n == 256 and (sname != 'ip' or pname == 'masked'):
                    continue
                cases.append((lname, loss, sname, scorer, pname, n, 'none'))
for method, nn in (('dns', [128, 64]), ('sir', [128, 64])):
    cases.append(('BPR', ra.BPRLoss, 'ip', ra.InnerProductScorer, 'uniform', nn, method))
base = None
for lname, loss, sname, scorer, pname, n, method in cases:
    sampler = {'uniform': lambda: ra.UniformSampler(N), 'popular': lambda: ra.PopularSamplerModel(counts),
               'masked': lambda: ra.MaskedUniformSampler(N)}[pname]()
    try:
        m = build(loss(), scorer(), sampler, n, method, excluding_hist=(pname == 'masked'))
        batch = {'user_id': torch.randint(1, U, (B,), device=dev), 'item_id': torch.randint(1, N, (B,), device=dev),
                 'rating': torch.ones(B, device=dev)}
        if pname == 'masked':
            batch['user_hist'] = hist[batch['user_id']]
        def fwd_bwd():
            m.zero_grad(set_to_none=True)
            m.training_step(batch).backward()
        t_f = T(lambda: m.training_step(batch))
        t_fb = T(fwd_bwd)
        nn = n if isinstance(n, int) else n[1]
        per = t_fb / (B * nn / 1e6)
        if base is None: base = per
        print(f'{lname:4s} {sname:4s} {pname:8s} n={str(n):10s} {method:5s}: step {t_f:7.3f} ms  +backward {t_fb:7.3f} ms  '
              f'{per:6.3f} ms/Mtriplet ({per / base:4.1f}x)', flush=True)
        del m
    except Exception as e:
        print(f'{lname} {sname} {pname} n={n} {method}: ERROR {repr(e)[:160]}', flush=True)
    torch.cuda.empty_cache()
