import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
from recstudio_amd import _native as nat
from bench import zipf_counts
dev = torch.device('cuda', 0)
N, d, b3, n3 = 1_000_001, 128, 8192, 256
item = torch.randn(N, d, device=dev) * 0.02
q3 = torch.randn(b3, d, device=dev) * 0.02
pos3 = torch.randint(1, N, (b3,), device=dev)
counts = zipf_counts(10_000_001, 100_000_000)
ps3 = ra.PopularSamplerModel(counts[:N]).to(dev)
kw3 = dict(pos_ids=pos3, sampler=nat.SAMPLER_POPULAR, **ps3.lookup_kwargs())
def T(fn, reps=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
buf = {}
def fwd():
    buf['o'] = ra.ops.fused_forward(item, q3, n3, out=buf.get('o'), **kw3)
print('lines_log2', ps3.lines_log2)
print('fwd popular n=256: %.4f ms' % T(fwd))
o = buf['o']
print('ssm loss:          %.4f ms' % T(lambda: ra.ops.pairwise_loss(nat.LOSS_SSM, o['pos_score'], o['neg_score'], o['pos_logp'], o['neg_logp'])))
print('bpr loss:          %.4f ms' % T(lambda: ra.ops.pairwise_loss(nat.LOSS_BPR, o['pos_score'], o['neg_score'])))
kwu = dict(pos_ids=pos3, sampler=nat.SAMPLER_UNIFORM)
bufu = {}
def fwdu():
    bufu['o'] = ra.ops.fused_forward(item, q3, n3, out=bufu.get('o'), **kwu)
print('fwd uniform n=256: %.4f ms' % T(fwdu))
