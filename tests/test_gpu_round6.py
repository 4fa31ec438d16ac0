"""Round 6 GPU parity tests: the full-softmax backward that never holds [B, N], float64 referees for the scatter gradients."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rel_close(a, b, rtol=1e-4, atol=1e-6):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


@pytest.fixture(scope='module')
def ra():
    import recstudio_amd
    recstudio_amd._native.lib()
    return recstudio_amd


@pytest.mark.parametrize('N,B,d', [(30_001, 256, 128), (5_003, 77, 64), (1_234, 33, 32), (40_007, 1000, 100), (130, 2048, 128), (60_001, 2048, 32)])
def test_full_softmax_backward_never_holds_the_score_matrix(ra, N, B, d, monkeypatch):
    """scorer.full_lse (SoftmaxLoss over the whole catalog: loss_func.py:39-47 over scorer.py:16) under autograd, the two forms that
    never write [B, N]:
      'flash' (default)  forward = rsa_fullscore_lse_grad (logsumexp AND softmax @ items in one pass, running reference with
                         rescaling), backward = upstream * that block + rsa_fullscore_softmax_dw (item-stationary recompute);
      'recompute'        lse-only forward, d/d query from the query-stationary recompute pass with the softmax tile NOT written
                         (rsa_fullscore_softmax_dq, probs = NULL), d/d items as above;
    both == float64 autograd (rtol 2e-4: fp32 sums of B resp. N terms), logsumexp to 1e-5, row 0 of the table gradient exactly
    zero, ragged sizes (partial item tiles, partial batch chunks, padded embed_dim), an arbitrary upstream gradient per row
    (sign and zero included), scores spread over +-30 so that the flash pass's reference moves -- and no [B, N-1] allocation:
    where that matrix would dwarf everything else (B = 2048, N = 60 001, d = 32: 492 MB against 7.7 MB of table) the peak
    memory of forward + backward stays below a QUARTER of its size.  == the stored-softmax form of round 5."""
    from recstudio_amd import scorer
    g = torch.Generator(device=DEV).manual_seed(N + B)
    w0 = torch.randn(N, d, device=DEV, generator=g) * 0.1
    w0[N // 2:] *= 6.0                                    # the later items score far higher: the running reference has to move
    q0 = torch.randn(B, d, device=DEV, generator=g) * (1.5 if d >= 64 else 3.0)
    up = torch.randn(B, device=DEV, generator=g)
    up[::7] = 0
    wd, qd = w0.double().requires_grad_(True), q0.double().requires_grad_(True)
    ref = torch.logsumexp(qd @ wd[1:].t(), -1)
    (ref * up.double()).sum().backward()
    smax = float((qd.detach() @ wd.detach()[1:].t()).abs().max())
    assert smax > 12.0
    # an fp32 score of magnitude s carries ~s * 2^-23 * sqrt(d)/2 of absolute error, which exp() turns into a RELATIVE error of the
    # softmax entry (any fp32 implementation, torch's included): the absolute tolerance follows the largest score
    tol = max(2e-6, 1e-6 * smax)
    scale_w, scale_q = tol / 2e-6 * float(wd.grad.abs().max()), tol / 2e-6 * float(qd.grad.abs().max())
    got = {}
    for mode in ('flash', 'recompute', 'store'):
        monkeypatch.setattr(scorer, 'FULL_SOFTMAX_BACKWARD', mode)
        w, q = w0.clone().requires_grad_(True), q0.clone().requires_grad_(True)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        lse = scorer.full_lse(q, w)
        (lse * up).sum().backward()
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated() - base
        torch.testing.assert_close(lse.detach().double(), ref.detach(), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(w.grad.double(), wd.grad, rtol=2e-4, atol=2e-6 * scale_w)
        torch.testing.assert_close(q.grad.double(), qd.grad, rtol=2e-4, atol=2e-6 * scale_q)
        assert bool((w.grad[0] == 0).all()), mode
        if mode != 'store' and B * (N - 1) * 4 > (256 << 20):
            assert peak < B * (N - 1) * 4 // 4, (mode, peak, B * (N - 1) * 4)
        got[mode] = (w.grad, q.grad)
    for mode in ('recompute', 'store'):
        torch.testing.assert_close(got[mode][0], got['flash'][0], rtol=2e-4, atol=2e-6 * scale_w)
        torch.testing.assert_close(got[mode][1], got['flash'][1], rtol=2e-4, atol=2e-6 * scale_q)
    # only the table needs a gradient: no query-gradient work at all (lse-only forward + the item-stationary pass)
    monkeypatch.setattr(scorer, 'FULL_SOFTMAX_BACKWARD', 'flash')
    w = w0.clone().requires_grad_(True)
    (scorer.full_lse(q0, w) * up).sum().backward()
    torch.testing.assert_close(w.grad, got['flash'][0], rtol=2e-4, atol=2e-6 * scale_w)      # (its lse comes from the lse-only kernel)


def test_flash_forward_entry_point(ra):
    """rsa_fullscore_lse_grad alone: lse and softmax @ items vs float64, an extreme catalog (one item 80 above the rest: the
    reference jumps, everything before it rescales to ~0), argument checks."""
    nat = ra._native
    N, B, d = 20_001, 130, 128
    g = torch.Generator(device=DEV).manual_seed(2)
    w = torch.randn(N, d, device=DEV, generator=g) * 0.1
    q = torch.randn(B, d, device=DEV, generator=g)
    w[N - 7] = q[5] * 80.0 / float(q[5] @ q[5])           # <q_5, w> = 80
    lse, gq = ra.ops.fullscore_lse_grad(w, q)
    S = q.double() @ w.double()[1:].t()
    torch.testing.assert_close(lse.double(), torch.logsumexp(S, -1), rtol=1e-5, atol=1e-5)
    want = torch.softmax(S, -1) @ w.double()[1:]
    torch.testing.assert_close(gq.double(), want, rtol=2e-4, atol=2e-6 * float(want.abs().max()))
    lib, p = nat.lib(), nat.ptr
    ws = torch.empty(int(lib.rsa_fullscore_lse_grad_workspace_bytes(B, N, d)), dtype=torch.uint8, device=DEV)
    assert lib.rsa_fullscore_lse_grad(p(w), N, d, p(q), B, p(lse), p(gq), p(ws), 16, None) == -1
    assert lib.rsa_fullscore_lse_grad(p(w), N, 96, p(q), B, p(lse), p(gq), p(ws), ws.numel(), None) == -3
    assert lib.rsa_fullscore_lse_grad(p(w), N, d, p(q), 0, p(lse), p(gq), None, 0, None) == 0


def test_full_softmax_dw_entry_point_checks(ra):
    nat = ra._native
    lib = nat.lib()
    w = torch.zeros(10, 128, device=DEV)
    q = torch.zeros(4, 128, device=DEV)
    lse = torch.zeros(4, device=DEV)
    out = torch.ones(10, 128, device=DEV)
    p = nat.ptr
    assert lib.rsa_fullscore_softmax_dw(p(w), 10, 128, p(q), 0, p(lse), None, p(out), None) == 0       # empty batch: all zero
    torch.cuda.synchronize()
    assert float(out.abs().sum()) == 0.0
    assert lib.rsa_fullscore_softmax_dw(p(w), 10, 96, p(q), 4, p(lse), None, p(out), None) == -3
    assert lib.rsa_fullscore_softmax_dw(None, 10, 128, p(q), 4, p(lse), None, p(out), None) == -1
    assert lib.rsa_fullscore_softmax_dw(p(w), 1, 128, p(q), 4, p(lse), None, p(out), None) == -1
    # uniform scores: softmax = 1 / (N - 1), d/d items = sum_b q_b / (N - 1)
    q = torch.randn(4, 128, device=DEV)
    lse = torch.full((4,), float(np.log(9.0)), device=DEV)
    got = ra.ops.fullscore_softmax_dw(w, q, lse)
    torch.testing.assert_close(got[1:], (q.sum(0) / 9.0).expand(9, 128), rtol=1e-5, atol=1e-6)
    assert float(got[0].abs().sum()) == 0.0


@pytest.mark.parametrize('loss', ['bpr', 'ssm'])
def test_scatter_gradients_float64_referee_at_headline_batch(ra, loss):
    """VERDICT r5 "next" #7: the row-sparse / sorted-scatter gradients of the fused training path are sums of up to hundreds of fp32
    terms in an order that differs from ATen's, held to rtol 2e-4 .. 3e-4 against the fp32 oracle by the suite.  Here the
    referee is float64 at the headline batch (B = 65 536, n = 64, d = 128; a 200 001-row catalog, so that a row collects ~21
    terms and the popular ones thousands): the kernel's item / user gradients are no further from float64 than torch's own fp32
    autograd on the device is (RMS error ratio <= 1.5, max error <= 2 x + one ulp of the largest gradient), and both sit inside
    rtol 3e-4 of it."""
    from recstudio_amd import fused
    N, U, d, B, n = 200_001, 50_001, 128, 65536, 64
    g = torch.Generator(device=DEV).manual_seed(17)
    iw = torch.randn(N, d, device=DEV, generator=g) * 0.1
    iw[0] = 0
    uw = torch.randn(U, d, device=DEV, generator=g) * 0.1
    uid = torch.randint(1, U, (B,), device=DEV, generator=g)
    pos = torch.randint(1, N, (B,), device=DEV, generator=g)
    sampler = ra.PopularSamplerModel((torch.rand(N) ** 6 * 5000).long() + 1).to(DEV)
    wi, wu = iw.clone().requires_grad_(True), uw.clone().requires_grad_(True)
    fn = fused.fused_bpr_loss if loss == 'bpr' else fused.fused_ssm_loss
    val, neg = fn(wi, wu, n, query_index=uid, pos_ids=pos, sampler=sampler)
    val.backward()

    def reference(dtype):
        a, b = iw.to(dtype).requires_grad_(True), uw.to(dtype).requires_grad_(True)
        qv = b[uid]
        ps = (qv * a[pos]).sum(-1)
        ns = torch.empty(B, n, dtype=dtype, device=DEV)
        chunk = 8192                                         # [chunk, n, d] gathered rows at a time (float64: 0.5 GB)
        total = 0
        for lo in range(0, B, chunk):
            sl = slice(lo, lo + chunk)
            nsc = (qv[sl].unsqueeze(1) * a[neg[sl]]).sum(-1)
            if loss == 'bpr':
                part = -torch.nn.functional.logsigmoid(ps[sl].unsqueeze(1) - nsc).mean(1).sum() / B
            else:
                lp = torch.log(sampler.pop_prob[pos[sl]]).to(dtype)
                ln = torch.log(sampler.pop_prob[neg[sl]]).to(dtype)
                zp, zn = ps[sl] - lp, nsc - ln
                part = (torch.logsumexp(torch.cat([zp.view(-1, 1), zn], 1), -1) - zp).sum() / B
            part.backward(retain_graph=True)
            total = total + part.detach()
        ga = a.grad.clone()
        ga[0] = 0                                            # nn.Embedding(padding_idx = 0)
        return total, ga, b.grad

    v64, gi64, gu64 = reference(torch.float64)
    v32, gi32, gu32 = reference(torch.float32)
    np.testing.assert_allclose(val.item(), v64.item(), rtol=1e-5)
    for got, g32, g64, name in ((wi.grad, gi32, gi64, 'item'), (wu.grad, gu32, gu64, 'user')):
        ek, eo = (got.double() - g64).abs(), (g32.double() - g64).abs()
        ulp = float(g64.abs().max()) * 2.0 ** -23
        rms_k, rms_o = float(ek.pow(2).mean().sqrt()), float(eo.pow(2).mean().sqrt())
        assert rms_k <= 1.5 * rms_o + 1e-12, (name, rms_k, rms_o)
        assert float(ek.max()) <= 2.0 * float(eo.max()) + ulp, (name, float(ek.max()), float(eo.max()), ulp)
        torch.testing.assert_close(got.double(), g64, rtol=3e-4, atol=4 * ulp)
        torch.testing.assert_close(g32.double(), g64, rtol=3e-4, atol=4 * ulp)


def test_bench_multi_rank_branch_runs_staged_world2():
    """VERDICT r5 "next" #9: the N > 1 branch of bench.py cannot rot -- two STAGED ranks on the one test GPU (RSA_BENCH_STAGED=1:
    collectives carried by gloo through host memory, tools/staged_dist.py) run the `--gpus 2` path at a small size through the
    launcher the driver uses and print ONE JSON line with the contract's keys, the in-job world-1 reference and the flat
    side figures inside `roofline`.  Exercises the code path and the keys; not a measurement."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RSA_BENCH_STAGED='1', RSA_BENCH_EXTRAS_DEADLINE_S='240', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
           '--items', '200001', '--users', '20001', '--batch', '512', '--neg', '64']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline'):
        assert key in line, key
    assert line['n_gpus'] == 2 and line['steps'] == 3 and line['scaling'] == 'weak' and line['value'] > 0
    assert line['config']['global_batch'] == 1024 and 'row-sharded x2' in line['config']['parallelism']
    roof = line['roofline']
    for key in ('frac', 'achieved', 'peak', 'world1_ms_per_step', 'efficiency_vs_world1', 'train_step_ms', 'train_step_ssm_ms'):
        assert isinstance(roof.get(key), (int, float)), (key, roof)
    assert line['backend'].startswith('staged') and line['ranks_seen'] == 2


@pytest.mark.parametrize('loss,n,sampler', [('bpr', 64, 'popular'), ('bpr', 1024, 'uniform'), ('ssm', 256, 'popular')])
def test_training_forward_without_the_score_store(ra, loss, n, sampler):
    """ops.fused_forward(want_scores=False) -- what the autograd training step issues: the kernel fuses the loss, so the [M, n]
    scores (4 B per triplet) are neither written nor returned (rsa_fused_args.neg_score = NULL); ids, loss, d loss/d score and
    d loss/d query are bit-equal to the call that keeps them.  Without a fused loss a null neg_score is an argument error."""
    nat = ra._native
    N, U, d, B = 50_001, 4_001, 128, 512
    g = torch.Generator(device=DEV).manual_seed(7)
    item = torch.randn(N, d, device=DEV, generator=g) * 0.1
    user = torch.randn(U, d, device=DEV, generator=g) * 0.1
    uid = torch.randint(1, U, (B,), device=DEV, generator=g)
    pos = torch.randint(1, N, (B,), device=DEV, generator=g)
    kw = dict(query_index=uid, pos_ids=pos, fused_loss=loss, want_query_grad=True)
    if sampler == 'popular':
        ps = ra.PopularSamplerModel(torch.randint(1, 50, (N,), generator=torch.Generator().manual_seed(1))).to(DEV)     # (no zero-probability positives: their loss is NaN, like the reference's)
        kw.update(sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs())
    else:
        kw.update(sampler=nat.SAMPLER_UNIFORM)
    torch.manual_seed(11)
    full = ra.ops.fused_forward(item, user, n, **kw)
    torch.manual_seed(11)
    lean = ra.ops.fused_forward(item, user, n, want_scores=False, **kw)
    assert 'neg_score' in full and 'neg_score' not in lean
    for k in ('neg_ids', 'loss', 'row_loss', 'dpos', 'dneg', 'query_grad', 'pos_score'):
        assert torch.equal(full[k], lean[k]), k
    a = nat.FusedArgs()
    a.item_table, a.n_items, a.dim, a.query, a.n_query_rows = nat.ptr(item), N, d, nat.ptr(user), U
    a.n_queries, a.num_neg, a.sampler, a.neg_ids = B, n, nat.SAMPLER_GIVEN, nat.ptr(full['neg_ids'])
    import ctypes
    assert nat.lib().rsa_fused_sample_gather_score(ctypes.byref(a), None) == -1


def test_full_softmax_training_step_config5_shape_properties(ra):
    """BASELINE.json configs[4] at full size through autograd (N = 1e6, d = 128, B = 2048; SoftmaxLoss = loss_func.py:39-47 over
    scorer.py:16), the default 'flash' form -- size-independent properties + float64 spot checks:
      * logsumexp == the lse-only kernel's (another code path) to 1e-5, and float64 on a sample of rows;
      * softmax rows sum to 1, so sum_i d loss/d item_i == sum_b g_b q_b (a checksum of the whole [N, d] gradient);
      * d loss/d item_0 == 0 (the padding row is not part of the catalog);
      * d loss/d query == float64 softmax @ items on a sample of rows; item-gradient rows == float64 on a sample of items;
      * the whole step peaks below 1 GB of extra memory where [B, N] alone would be 8 GB."""
    from recstudio_amd import scorer
    assert scorer.FULL_SOFTMAX_BACKWARD == 'flash'
    N, d, B = 1_000_001, 128, 2048
    g = torch.Generator(device=DEV).manual_seed(55)
    w = (torch.randn(N, d, device=DEV, generator=g) * 0.1).requires_grad_(True)
    q = (torch.randn(B, d, device=DEV, generator=g) * 0.5).requires_grad_(True)
    up = torch.rand(B, device=DEV, generator=g) / B
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    lse = scorer.full_lse(q, w)
    (lse * up).sum().backward()
    torch.cuda.synchronize()
    assert torch.cuda.max_memory_allocated() - base < (1 << 30)
    lse2 = ra.ops.fullscore(w.detach(), q.detach(), want_lse=True)[1]
    torch.testing.assert_close(lse.detach(), lse2, rtol=1e-5, atol=1e-5)
    want_sum = (up.double()[:, None] * q.detach().double()).sum(0)
    torch.testing.assert_close(w.grad.double().sum(0), want_sum, rtol=1e-4, atol=1e-6 * float(want_sum.abs().max()))
    assert bool((w.grad[0] == 0).all())
    rows = torch.tensor([0, 1, 1023, 2047], device=DEV)
    wd = w.detach().double()
    S = q.detach()[rows].double() @ wd[1:].t()
    torch.testing.assert_close(lse.detach()[rows].double(), torch.logsumexp(S, -1), rtol=1e-5, atol=1e-5)
    gq = up[rows].double()[:, None] * (torch.softmax(S, -1) @ wd[1:])
    torch.testing.assert_close(q.grad[rows].double(), gq, rtol=2e-4, atol=2e-6 * float(gq.abs().max()))
    items = torch.tensor([1, 2, 500_000, N - 1], device=DEV)
    P = torch.exp(q.detach().double() @ wd[items].t() - lse.detach().double()[:, None]) * up.double()[:, None]      # [B, 4]
    gi = P.t() @ q.detach().double()
    torch.testing.assert_close(w.grad[items].double(), gi, rtol=2e-4, atol=2e-6 * float(gi.abs().max()))


def test_lookahead_tickets_do_not_grow_the_allocator():
    """ShardedRetriever.prepare_step / training_step(ticket=) issued by a host that runs far ahead of the GPU (what a training loop
    that keeps its losses on the device does): a ticket's buffers are allocated on the second stream and freed after the main
    stream consumed them, so without a bound on the host's run-ahead every ticket is a fresh device allocation (15 -> 73 GB
    reserved in 1.5 s at the configs[3] shape).  ShardedItemTable.MAX_TICKETS_AHEAD bounds it: 600 look-ahead steps reserve no
    more than a few steps' buffers on top of the first ones."""
    import socket
    import torch.distributed as dist
    import recstudio_amd as ra
    from recstudio_amd.shard import RowShardPlan, ShardedItemTable, ShardedRetriever
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(s.getsockname()[1])
    s.close()
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        N, U, d, B, n = 400_001, 5000, 128, 2048, 256
        g = torch.Generator(device=DEV).manual_seed(3)
        item = torch.randn(N, d, device=DEV, generator=g) * 0.1
        tower = torch.nn.Embedding(U, d).to(DEV)
        uid = torch.randint(1, U, (B,), device=DEV, generator=g)
        pos = torch.randint(1, N, (B,), device=DEV, generator=g)
        table = ShardedItemTable(item, RowShardPlan(N, 1), 0, dist, check_every=0)
        tr = ShardedRetriever(table, tower, ra.UniformSampler(N), ra.BPRLoss(), n, item_sgd_lr=1e-3, query_sgd_lr=1e-3)
        assert tr.can_prepare()
        tr.training_step(uid, pos)
        tk = tr.prepare_step(uid, pos)
        for _ in range(8):
            nxt = tr.prepare_step(uid, pos)
            tr.training_step(uid, pos, ticket=tk)
            tk = nxt
        torch.cuda.synchronize()
        before = torch.cuda.memory_reserved()
        for _ in range(600):
            nxt = tr.prepare_step(uid, pos)
            tr.training_step(uid, pos, ticket=tk)
            tk = nxt
        grown = torch.cuda.memory_reserved() - before
        tr.training_step(uid, pos, ticket=tk)
        torch.cuda.synchronize()
        table.check_overflow()
        assert grown < (256 << 20), grown          # (unbounded: ~10 MB per step here, 6 GB over the loop)
    finally:
        dist.destroy_process_group()


def test_sgd_step_headline_shape_properties(ra):
    """The in-place SGD step (rsa_bpr_sgd_prepare / _apply: negatives drawn inside the first launch of ONE sort over the item AND
    the user elements, ABI 11) at BASELINE configs[1]'s own size -- N = 1e7 + 1, U = 1e6 + 1, B = 65 536, n = 64, popularity
    sampler: the negatives == torch's stream (``searchsorted(table, rand)``, sampler.py:246-247), loss == the plain forward's on the
    same weights, every touched row moved and no other, rows with ONE element == row - lr * d * q exactly as the all-sorted form
    computes it, user rows == the all-sorted form's bit for bit (same sorted order, same chunking), run-to-run bit equality."""
    N, U, d, B, n, lr = 10_000_001, 1_000_001, 128, 65536, 64, 3000.0      # (BPRLoss is a mean over B * n terms: lr ~ 0.05 per sample)
    g = torch.Generator(device=DEV).manual_seed(11)
    iw0 = torch.empty(N, d, device=DEV).normal_(0, 0.1, generator=g)
    iw0[0] = 0
    uw0 = torch.empty(U, d, device=DEV).normal_(0, 0.1, generator=g)
    counts = (torch.rand(N, generator=torch.Generator().manual_seed(2)) ** 8 * 1e4).long()
    ps = ra.PopularSamplerModel(counts).to(DEV)
    uid = torch.randint(1, U, (B,), device=DEV, generator=g)
    uid[:64] = uid[64:128]                         # users that occur twice in the batch
    pos = torch.randint(1, N, (B,), device=DEV, generator=g)
    res = []
    for mode in (True, True, False):
        iw, uw = iw0.clone(), uw0.clone()
        torch.manual_seed(5)
        loss, ids = ra.fused.bpr_sgd_step(iw, uw, n, lr, user_ids=uid, pos_ids=pos, sampler=ps, in_forward=mode)
        res.append((loss.clone(), ids.clone(), iw, uw))
        torch.cuda.synchronize()
    (l1, i1, it1, us1), (l2, i2, it2, us2), (l0, i0, it0, us0) = res
    torch.manual_seed(5)
    want = torch.searchsorted(ps.table, torch.rand(B, n, device=DEV)).clamp_(max=N - 1)
    assert torch.equal(i1, want) and torch.equal(i0, want)
    assert torch.equal(l1, l2) and torch.equal(it1, it2) and torch.equal(us1, us2)          # run to run
    assert torch.equal(l1, l0) and torch.equal(us1, us0)
    torch.manual_seed(5)
    o = ra.ops.fused_forward(iw0, uw0, n, query_index=uid, pos_ids=pos, sampler=ra._native.SAMPLER_POPULAR, fused_bpr=True,
                             **ps.lookup_kwargs())
    assert torch.equal(o['neg_ids'], i1)
    rel_close(l1.cpu(), o['loss'].cpu(), rtol=1e-6)
    # which rows moved: exactly the touched ones (padding row 0 never)
    touched = torch.zeros(N, dtype=torch.bool, device=DEV)
    touched[pos] = True
    touched[i1.reshape(-1)] = True
    touched[0] = False
    moved = (it1 != iw0).any(1)
    assert not bool((moved & ~touched).any()) and not it1[0].any()
    assert int((touched & ~moved).sum()) <= 8          # (a gradient that rounds to nothing on every component: practically never)
    cnt = torch.bincount(torch.cat([pos, i1.reshape(-1)]), minlength=N)
    solo = cnt == 1
    solo[0] = False
    assert 0.3 < float(solo[i1.reshape(-1)].float().mean()) < 0.7
    assert torch.equal(it1[solo], it0[solo])            # solo rows: the apply pass's arithmetic, rounding for rounding
    shared = touched & ~solo
    rel_close(it1[shared].cpu(), it0[shared].cpu(), rtol=1e-5, atol=1e-7)
    um = torch.zeros(U, dtype=torch.bool, device=DEV)
    um[uid] = True
    um[0] = False
    assert torch.equal((us1 != uw0).any(1), um)
