"""One bench figure, alone, for rocprofv3 (tools/collect_shapes.sh): builds the shape, runs K launches of its step
and prints {"shape", "steps"} so that the summariser can turn totals into per-step figures.
usage: python tools/prof_shapes.py SHAPE [K]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recstudio_amd as ra                      # noqa: E402
from recstudio_amd import _native as nat        # noqa: E402
from bench import zipf_counts                   # noqa: E402

shape = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
d, U = 128, 1_000_001
gen = torch.Generator(device=dev).manual_seed(100)


def popular_sampler(n_items, tag):
    """The popularity tables of an N-item Zipf catalog; cached under /tmp across the passes of one collection run (the
    host-side build of the 1e8-item tables takes minutes)."""
    path = f'/tmp/rsa_ps_{tag}.pt'
    if os.path.exists(path):
        return torch.load(path, weights_only=False).to(dev)
    ps = ra.PopularSamplerModel(zipf_counts(n_items, 100_000_000))
    torch.save(ps, path)
    return ps.to(dev)


def table(n_items, seed):
    t = torch.empty(n_items, d, device=dev).normal_(0, 0.02, generator=torch.Generator(device=dev).manual_seed(seed))
    t[0] = 0
    return t


user = torch.empty(U, d, device=dev).normal_(0, 0.02, generator=torch.Generator(device=dev).manual_seed(3))
torch.manual_seed(2022)
buf = {}
if shape in ('headline_N1e7_popular_n64_B65536', 'N1e7_popular_n64_B4096', 'N1e7_popular_n64_B16384'):
    B = int(shape.rsplit('_B', 1)[1])
    item = table(10_000_001, 1)
    ps = popular_sampler(10_000_001, '1e7')
    uid = torch.randint(1, U, (B,), device=dev, generator=gen)
    pos = torch.randint(1, 10_000_001, (B,), device=dev, generator=gen)
    kw = dict(query_index=uid, pos_ids=pos, sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs())

    def step():
        buf['o'] = ra.ops.fused_forward(item, user, 64, out=buf.get('o'), fused_bpr=True, **kw)
elif shape in ('queue_N1e7_popular_n64_B4096x16', 'queue_N1e7_popular_n64_B16384x4'):
    # S independent batches consumed by ONE launch (rsa_fused_args.n_batches): per-batch time = kernel time / S
    bs = shape.rsplit('_B', 1)[1]
    B, S = (int(v) for v in bs.split('x'))
    item = table(10_000_001, 1)
    ps = popular_sampler(10_000_001, '1e7')
    uid = torch.randint(1, U, (S * B,), device=dev, generator=gen)
    pos = torch.randint(1, 10_000_001, (S * B,), device=dev, generator=gen)
    kw = dict(query_index=uid, pos_ids=pos, sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs())

    def step():       # (through fused_forward, not FusedStep: 35 us of host time per call do not matter under a 0.45 ms launch)
        buf['o'] = ra.ops.fused_forward(item, user, 64, out=buf.get('o'), fused_bpr=True, n_batches=S, **kw)
elif shape.startswith('N1e8_'):
    n8 = 100_000_001
    item = table(n8, 8)
    _, smp, nn, bb = shape.split('_')
    n_neg, B = int(nn[1:]), int(bb[1:])
    uid = torch.randint(1, U, (B,), device=dev, generator=gen)
    pos = torch.randint(1, n8, (B,), device=dev, generator=gen)
    if smp == 'popular':
        ps = popular_sampler(n8, '1e8')
        kw = dict(query_index=uid, pos_ids=pos, sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs())
    else:
        kw = dict(query_index=uid, pos_ids=pos, sampler=nat.SAMPLER_UNIFORM)

    def step():
        buf['o'] = ra.ops.fused_forward(item, user, n_neg, out=buf.get('o'), fused_bpr=True, want_mean=False, **kw)
elif shape == 'ssm_N1e6_popular_n256_B8192':
    n6, B, n_neg = 1_000_001, 8192, 256
    item = table(n6, 1)
    ps = popular_sampler(n6, '1e6')
    q = user[1:B + 1].contiguous()
    pos = torch.randint(1, n6, (B,), device=dev, generator=gen)
    kw = dict(pos_ids=pos, sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs())

    def step():
        buf['o'] = ra.ops.fused_forward(item, q, n_neg, out=buf.get('o'), fused_loss='ssm', **kw)
elif shape in ('sharded_world1_step', 'sharded_world1_train', 'sharded_world1_train_ssm', 'sharded_world1_train_nondet'):
    import torch.distributed as dist
    from recstudio_amd import shard
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29578')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    n_blk, n_neg, B = 12_500_001, 1024, 4096
    item = table(n_blk, 9)
    tbl = shard.ShardedItemTable(item, shard.RowShardPlan(n_blk, 1), 0, dist, check_every=0, deterministic=not shape.endswith('_nondet'))      # (deterministic router: the default since round 6)
    us = ra.UniformSampler(n_blk)
    uid = torch.randint(1, U, (B,), device=dev, generator=gen)
    pos = torch.randint(1, n_blk, (B,), device=dev, generator=gen)
    if shape == 'sharded_world1_step':
        def step():
            tbl.sample_and_score(user, uid, pos, n_neg, us, fused_loss='bpr', want_ids=False, want_grad=True)
    else:
        tower = torch.nn.Embedding(U, d).to(dev)
        # one stream: the tracked figure is a SUM of kernel durations (with the query rows on the second stream, the default,
        # the kernels that run side by side would each be charged the whole overlap)
        loss_fn = ra.SampledSoftmaxLoss() if shape.endswith('_ssm') else ra.BPRLoss()      # (_ssm: on the owners, two phases)
        trainer = shard.ShardedRetriever(tbl, tower, us, loss_fn, n_neg, item_sgd_lr=0.05, query_sgd_lr=0.05,
                                         overlap_query_rows=False)

        def step():
            trainer.training_step(uid, pos)
elif shape in ('sgd_step_N1e7_popular_n64_B65536', 'adam_step_N1e7_popular_n64_B65536'):
    from recstudio_amd.fused import FusedBPRAdam, bpr_sgd_step
    B = 65536
    item = table(10_000_001, 1)
    ps = popular_sampler(10_000_001, '1e7')
    uid = torch.randint(1, U, (B,), device=dev, generator=gen)
    pos = torch.randint(1, 10_000_001, (B,), device=dev, generator=gen)
    if shape.startswith('sgd'):
        def step():
            bpr_sgd_step(item, user, 64, 1e-3, user_ids=uid, pos_ids=pos, sampler=ps)
    else:
        fa = FusedBPRAdam(item, user, lr=1e-3)

        def step():
            fa.step(64, user_ids=uid, pos_ids=pos, sampler=ps)
elif shape == 'train_step_N1e7_popular_n64_B65536':
    B = 65536
    item = table(10_000_001, 1)
    ps = popular_sampler(10_000_001, '1e7')
    uid = torch.randint(1, U, (B,), device=dev, generator=gen)
    pos = torch.randint(1, 10_000_001, (B,), device=dev, generator=gen)
    kw = dict(query_index=uid, pos_ids=pos, sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs())

    def step():       # forward with the user gradient + write-only row-sparse item-gradient rows (bench.py train_step)
        buf['o'] = ra.ops.fused_forward(item, user, 64, out=buf.get('o'), fused_bpr=True, want_query_grad=True, want_scores=False, **kw)
        o = buf['o']
        ra.ops.fused_backward(item, user, o['neg_ids'], o['dneg'], query_index=uid, pos_ids=pos, dpos=o['dpos'],
                              dense_item_grad=False, row_item_grad=True, want_query_grad=False)
elif shape in ('fullscore_lse_B2048_N1e6', 'fullscore_top100_B2048_N1e6'):
    item = table(1_000_001, 1)
    q = user[1:2049].contiguous()
    k = 100 if 'top100' in shape else 0

    def step():
        ra.ops.fullscore(item, q, want_lse=True, k=k)
elif shape == 'softmax_dx_B2048_N1e6':
    # d loss/d items of the full softmax: probs^T @ query, item-stationary on the fp32 MFMA (rsa_probs_t_query)
    item = table(1_000_001, 1)
    q = user[1:2049].contiguous()
    lse = ra.ops.fullscore(item, q, want_lse=True)[1]
    probs = ra.ops.fullscore_softmax(item, q, lse, torch.full((2048,), 1.0 / 2048, device=dev))
    gx = torch.empty(1_000_000, d, device=dev)

    def step():
        ra.ops.probs_t_query(probs, q, out=gx)
elif shape in ('softmax_dw_B2048_N1e6', 'softmax_dq_nowrite_B2048_N1e6', 'softmax_train_B2048_N1e6', 'softmax_flash_fwd_B2048_N1e6'):
    # the full-softmax backward that never holds [B, N] (round 6): d/d items with the softmax tile recomputed in registers
    # (rsa_fullscore_softmax_dw); d/d query from the recompute pass without the score store; the whole training step
    item = table(1_000_001, 1)
    q = user[1:2049].contiguous()
    lse = ra.ops.fullscore(item, q, want_lse=True)[1]
    sc = torch.full((2048,), 1.0 / 2048, device=dev)
    gw = torch.empty_like(item)
    if shape.startswith('softmax_dw'):
        def step():
            ra.ops.fullscore_softmax_dw(item, q, lse, sc, out=gw)
    elif shape.startswith('softmax_dq'):
        def step():
            ra.ops.fullscore_softmax(item, q, lse, sc, want_query_grad=True, want_probs=False)
    elif shape.startswith('softmax_flash'):
        def step():       # the flash forward: logsumexp + d lse/d query in one pass (rsa_fullscore_lse_grad)
            ra.ops.fullscore_lse_grad(item, q)
    else:
        from recstudio_amd.scorer import full_lse
        wt, qt = item.requires_grad_(True), q.clone().requires_grad_(True)

        def step():
            wt.grad = qt.grad = None
            full_lse(qt, wt).mean().backward()
elif shape == 'seg_gather_B8192_L50':
    n6, B, L = 1_000_001, 8192, 50
    item = table(n6, 1)
    lens = torch.randint(1, L + 1, (B,), device=dev, generator=gen)
    end = torch.cumsum(lens, 0)
    start = end - lens
    flat = torch.randint(1, n6, (int(end[-1]),), device=dev, generator=gen)

    def step():       # the call SASRecQueryEncoder makes on the device loader's CSR view: rows only
        ra.ops.seg_gather(item, flat, start, end, L, want_rows=True, want_ids=False)
else:
    raise SystemExit(f'unknown shape {shape}')

# Warm-up by TIME, not by count: a 0.4 ms step launched 20 times is 8 ms of GPU work, not enough for the clocks to settle
# (the same kernel measured 7 % slower this way than inside bench.py's 250-launch run on the same box).  WARM_MS of
# back-to-back launches first (counted, so that the summariser can skip them), then the K counted ones.
import time                                     # noqa: E402
# (round 5: 120 ms was not enough either -- the chip needs about a SECOND of load after idling before the same launch reaches
# its steady time, profiles/r05_warmup_ramp.json: rounds 1-4's tracked profiles were 7-11 % "cold")
WARM_MS = float(os.environ.get('PROF_WARM_MS', '2500'))
WARM = 0
for _ in range(3):
    step()
    WARM += 1
torch.cuda.synchronize()
t0 = time.perf_counter()
while (time.perf_counter() - t0) * 1e3 < WARM_MS:
    for _ in range(5):
        step()
        WARM += 1
    torch.cuda.synchronize()
# Shapes whose step is ONE launch writing into buf['o']: the same launch is 10-15 % slower on some allocations of its output
# arrays than on others (bimodal, per allocation, cause unknown: DESIGN 6, profiles/r05_output_placement.json) -- and a fresh
# process that allocates tables, then outputs, lands on such an allocation more often than bench.py's long-lived one.  The
# tracked figure is the KERNEL's: PROF_OUT_SETS (default 4) output sets are allocated (all kept alive), each timed with events,
# the counted launches run on the fastest; every set's time is printed.
info = {}
n_sets = int(os.environ.get('PROF_OUT_SETS', '4'))
if 'o' in buf and n_sets > 1:
    sets = [buf['o']]
    for _ in range(n_sets - 1):
        buf.pop('o')
        step()
        WARM += 1
        sets.append(buf['o'])
    times = []
    for o in sets:
        buf['o'] = o
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) < 0.3:
            for _ in range(5):
                step()
                WARM += 1
            torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        for _ in range(30):
            step()
            WARM += 1
        a1.record()
        torch.cuda.synchronize()
        times.append(round(a0.elapsed_time(a1) / 30 * 1e3, 2))
    best = min(range(len(sets)), key=lambda i: times[i])
    buf['o'] = sets[best]
    info = {'out_set_us': times, 'out_set_used': best}
    for _ in range(20):
        step()
        WARM += 1
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(K):
    step()
e1.record()
torch.cuda.synchronize()
print(json.dumps(dict({'shape': shape, 'steps': K, 'warmup': WARM, 'event_us_per_step': round(e0.elapsed_time(e1) / K * 1e3, 2)}, **info)))
