#!/usr/bin/env python
"""bench.py -- M scored (user, pos, neg) triplets / s at d = 128 on MI355X.

One *step* = one pass of the hot path over one batch of synthetic interactions:
    fused [user-row gather + popularity sampling (in-kernel Philox + inverse CDF) + negative-row
    gather + inner-product scoring]  ->  BPR loss (+ d loss / d score)
i.e. BaseRetriever.forward + loss_fn of the reference (baseretriever.py:142-171, loss_func.py:55-59)
on BASELINE.json configs[1]: N = 10 000 001 items x d = 128 fp32, 1 000 001 users, popularity
sampler, n = 64 negatives, inner product, BPR.  All inputs are resident in HBM before the timed
region.

--gpus N > 1: BASELINE.json configs[3] -- a 100 000 001-item table row-sharded over the N ranks, n = 1024
negatives, B = 4096 queries per GPU per step (weak scaling), ids out / scores back by RCCL all-to-all.  Started
without a launcher (WORLD_SIZE unset) the script re-executes itself under torch.distributed.run with N ranks;
under the driver's own torchrun it checks that the world size is N.

Prints ONE JSON line (rank 0).  Extra keys beyond the driver contract: roofline, cpu_baseline,
train_step (forward + loss + row-sparse gradient scatter), sweep.
"""
import argparse
import threading
import json
import os
import sys
import time

# (the hosts' driver only supports dmabuf IPC: without this RCCL's cross-process handles fail with `hipIpcGetMemHandle: invalid
# argument`; exported on the GPU boxes already, set here as well so that a bare `torchrun bench.py` cannot miss it)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)


def bytes_per_triplet(d, n, popular, fused_loss=False):
    """negative row + (user row + positive row + two ids) / n + id/score/logp
    written + sampler probe; the fused BPR epilogue additionally writes d loss/d score (4 B per triplet) and
    the per-query loss and d loss/d pos (8 B per query)."""
    return 4 * d + 2 * 4 * d / n + 16.0 / n + 16 + (8 if popular else 0) + ((4 + 8.0 / n) if fused_loss else 0)


def make_workload(dev, n_items, n_users, d, seed=1):
    g = torch.Generator(device=dev).manual_seed(seed)
    item = torch.empty(n_items, d, device=dev).normal_(0, 0.02, generator=g)     # normal_initialization, init.py:18-27
    item[0] = 0
    user = torch.empty(n_users, d, device=dev).normal_(0, 0.02, generator=g)
    user[0] = 0
    return item, user


def zipf_counts(n_items, n_inter, seed=1):
    """bincount of a Zipf(alpha=1) item stream over a permuted id space, built on the CPU like
    TripletDataset.item_freq (dataset.py:1216-1230); sampled as expected counts + Poisson noise so
    that 1e8 interactions do not have to be materialised."""
    g = torch.Generator().manual_seed(seed)
    rank = torch.arange(1, n_items, dtype=torch.float64)
    p = 1.0 / rank
    p /= p.sum()
    lam = (p * n_inter).to(torch.float32)
    cnt = torch.poisson(lam, generator=g).long()
    perm = torch.randperm(n_items - 1, generator=g)
    counts = torch.zeros(n_items, dtype=torch.int64)
    counts[1:] = cnt[perm]
    return counts


PREWARM_S = 1.5


def prewarm(fn, seconds=None):
    """Keep the GPU busy with `fn` for `seconds` before a measurement.  Found in round 5 (profiles/r05_warmup_ramp.json, docs/HISTORY.md 6): after the
    chip has idled -- a fresh process, a host-side table build -- the SAME launch on the SAME allocation runs 7-11 % slower for
    about a second (448 -> 431 at 0.5 s -> 419 us from 1 s on; back to 449 after 5 s of idling) while rocm-smi / amd-smi report the
    same sclk / mclk / fclk.  Rounds 1-4 read that as box-to-box spread: the tracked profiles (120 ms of warm-up in a fresh
    process) were the cold number, the bench line (measured after half a minute of other figures) the warm one."""
    seconds = PREWARM_S if seconds is None else seconds
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()


def time_gpu(fn, steps, warmup, dist=None, warm_s=None):
    if dist is None:          # (time-based: ranks of a job would run different numbers of collective-bearing steps)
        prewarm(fn, warm_s)
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def time_gpu_best(fn, steps, warmup, repeats=3):
    """Secondary figures: best of a few short repeats (a single 50-step window on a shared box occasionally
    lands on a transient stall; the headline line keeps the contract's single K-step window)."""
    return min(time_gpu(fn, steps, warmup if r == 0 else 1, warm_s=None if r == 0 else 0.0) for r in range(repeats))


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run ... bench.py --gpus N`
    (one rank per GPU over RCCL, rendezvous on 127.0.0.1)."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n_gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.execvpe(cmd[0], cmd, env)


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _median_time(step, budget_s, min_steps=3, max_steps=30, warm=1):
    for _ in range(warm):
        step()
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < min_steps or (time.perf_counter() < t_end and len(times) < max_steps):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    times.sort()
    return times[len(times) // 2], len(times)


def cpu_baseline(args, counts, d, B, n):
    """SURVEY.md 8(d), CPU leg: the oracle (a port of the reference's PyTorch path, golden-checked against it; not the
    product) timed on this box's host cores.  (i) forward-only sample + gather + score + loss under no_grad at the
    headline's N / d / B / n, for 8 threads and for all cores (torch-CPU gathers do not scale to hundreds of threads: both
    are reported, `value` is the better one); (ii) the full training step with the reference's dense backward
    (autograd -> [N, d] gradients) at N = 1e6, the largest size where that is feasible, 3 warm-up + >= 5 timed steps."""
    import oracle
    nproc = os.cpu_count() or 1
    n_items = counts.numel()
    g = torch.Generator().manual_seed(1)
    t0 = time.perf_counter()
    torch.set_num_threads(nproc)
    item = torch.empty(n_items, d).uniform_(-0.035, 0.035, generator=g)
    item[0] = 0
    user = torch.empty(args.users, d).uniform_(-0.035, 0.035, generator=g)
    ps = oracle.PopularSamplerModel(counts)
    setup = time.perf_counter() - t0
    uid = torch.randint(1, args.users, (B,), generator=g)
    pos = torch.randint(1, n_items, (B,), generator=g)

    def step():
        with torch.no_grad():
            q = user[uid]
            lpp, neg, lnp = ps.forward(q, n, pos)
            p, s = oracle.retriever_forward(item, q, pos, neg)
            return oracle.bpr_loss(p, s)
    by_threads = {}
    for thr in sorted({min(nproc, t) for t in (8, 32, nproc)}):
        torch.set_num_threads(thr)
        med, cnt = _median_time(step, 6.0)
        by_threads[thr] = (med, cnt)
    best_thr = min(by_threads, key=lambda t: by_threads[t][0])
    med, cnt = by_threads[best_thr]
    res = {'value': round(B * n / med / 1e6, 3), 'unit': 'M triplets/s', 'cores': best_thr, 'kind': 'port',
           'nproc': nproc, 'cpu_model': _cpu_model(), 'torch': torch.__version__,
           'forward_ms_by_threads': {str(t): round(v[0] * 1e3, 1) for t, v in by_threads.items()},
           'sample': f'(i) oracle forward (popular sample + gather + inner product + BPR, no_grad) on torch-CPU, N={n_items} d={d} '
                     f'B={B} n={n} (the headline shape), median of {cnt} steps at {best_thr} threads = {med * 1e3:.1f} ms/step '
                     f'(table setup {setup:.1f}s untimed)'}
    # (ii) full step, dense backward, N = 1e6 (dense gradients are infeasible beyond)
    try:
        n2, b2 = 1_000_001, 4096
        item2 = item[:n2].clone()
        ps2 = oracle.PopularSamplerModel(counts[:n2])
        uid2, pos2 = uid[:b2], pos[:b2] % (n2 - 1) + 1
        torch.set_num_threads(best_thr)

        iw2, uw2 = item2.requires_grad_(True), user.clone().requires_grad_(True)
        emb = torch.nn.functional.embedding

        def train():          # recommender.py:636-639 on the oracle's ops: zero_grad (set to None), forward, loss, backward
            iw2.grad = uw2.grad = None
            q = emb(uid2, uw2, padding_idx=0)
            with torch.no_grad():
                _, neg, _ = ps2.forward(q, n, pos2)
            p_s = oracle.inner_product_score(q, emb(pos2, iw2, padding_idx=0))
            n_s = oracle.inner_product_score(q, emb(neg, iw2, padding_idx=0))
            oracle.bpr_loss(p_s, n_s).backward()
        for _ in range(2):
            train()
        med2, cnt2 = _median_time(train, 8.0, min_steps=5, max_steps=12, warm=1)
        res['train_step_dense_backward'] = {
            'ms_per_step': round(med2 * 1e3, 1), 'M_triplets_s': round(b2 * n / med2 / 1e6, 3), 'threads': best_thr,
            'what': f'(ii) sample + forward + BPR + autograd dense backward ([N, d] item and user gradients), N={n2} B={b2} '
                    f'n={n}, 3 warm-up + {cnt2} timed steps, median'}
    except Exception as e:
        res['train_step_dense_backward'] = {'error': repr(e)[:200]}
    torch.set_num_threads(nproc)
    return res


def _smi_json(*flags):
    import subprocess
    try:
        out = subprocess.run(['rocm-smi', *flags, '--json'], capture_output=True, text=True, timeout=20).stdout
        return json.loads(out[out.index('{'):])
    except Exception:
        return {}


def box_state(busy_fn=None, device=0):
    """The state of the box this process measured on (two boxes ran the same binary 10 % apart and
    nothing said why): engine / memory / fabric clocks and socket power sampled by rocm-smi WHILE `busy_fn` keeps the GPU
    busy (a thread launches it back to back for the duration of the query), the power cap, the performance level and the
    compute / memory partition modes.  Numbers only; {} when rocm-smi is missing."""
    stop = threading.Event()
    th = None
    if busy_fn is not None:
        def spin():
            torch.cuda.set_device(device)
            while not stop.is_set():
                for _ in range(50):
                    busy_fn()
                torch.cuda.synchronize()
        th = threading.Thread(target=spin, daemon=True)
        th.start()
        time.sleep(0.5)
    try:
        load = _smi_json('--showclocks', '--showpower', '--showtemp')
    finally:
        stop.set()
        if th is not None:
            th.join()
    static = _smi_json('--showmaxpower', '--showperflevel', '--showcomputepartition', '--showmemorypartition')
    card = next(iter(load.values()), {}) if load else {}
    card.update(next(iter(static.values()), {}) if static else {})
    import re
    out = {}

    def num(v):
        m = re.search(r'(-?\d+(?:\.\d+)?)', str(v))
        return float(m.group(1)) if m else None
    for key, val in card.items():
        k = key.lower()
        if 'sclk clock speed' in k:
            out['sclk_mhz'] = num(val)
        elif 'mclk clock speed' in k:
            out['mclk_mhz'] = num(val)
        elif 'fclk clock speed' in k:
            out['fclk_mhz'] = num(val)
        elif 'max graphics package power' in k:
            out['power_cap_w'] = num(val)
        elif 'package power' in k or 'socket power' in k:
            out['power_w'] = num(val)
        elif 'performance level' in k:
            out['perf_level'] = str(val)
        elif 'compute partition' in k:
            out['compute_partition'] = str(val)
        elif 'memory partition' in k:
            out['memory_partition'] = str(val)
        elif 'temperature (sensor junction)' in k:
            out['temp_junction_c'] = num(val)
        elif 'temperature (sensor hbm' in k or 'temperature (sensor memory)' in k:
            out['temp_hbm_c'] = max(out.get('temp_hbm_c') or 0.0, num(val) or 0.0)
    return out


PROFILE_ROUNDS = ('r06', 'r05', 'r04', 'r03')


def profile_record(name):
    """The committed rocprofv3 record of a bench figure (profiles/<round>_kernel_profiles.json, written by
    tools/collect_shapes.sh on a GPU box; the newest round that has the figure): kernel average from --kernel-trace --stats
    and the FETCH_SIZE / WRITE_SIZE passes.  bench.py prints `profile_frac` next to every live fraction that has one, so
    that a reader sees the tracked number and the live one side by side (boxes of the pool differ by several per cent)."""
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, 'profiles', f'{rnd}_kernel_profiles.json')
        try:
            rec = json.load(open(path)).get(name)
        except (OSError, ValueError):
            rec = None
        if rec:
            return dict(rec, _file=f'profiles/{rnd}_kernel_profiles.json')
    return None


def with_profile(entry, name, alg_bytes, whole_step=False, flops=None):
    """Attach the tracked profile of figure `name`: `profile_frac` = the same algorithmic bytes (or flops) over the tracked
    rocprofv3 time -- the dominant kernel's average, or with `whole_step` the sum of the step's kernels."""
    rec = profile_record(name)
    us = rec and rec.get('step_kernel_us' if whole_step else 'avg_us')
    if us:
        if flops is not None:
            entry['profile_frac'] = round(flops / (us * 1e-6) / 1e12 / 157.3, 4)
        else:
            entry['profile_frac'] = round(alg_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        entry['profile_step_kernel_ms' if whole_step else 'profile_avg_kernel_ms'] = round(us / 1e3, 4)
        if whole_step:
            entry['profile_dominant_kernel_share'] = rec.get('dominant_share_of_step')
        elif rec.get('hbm_bytes_per_launch') and alg_bytes:
            entry['profile_traffic_over_alg'] = round(rec['hbm_bytes_per_launch'] / alg_bytes, 3)
        entry['profile_source'] = f'{rec["_file"]}["{name}"]'
    return entry


def side_figures(extra):
    """Compact copy (numbers only) of the side figures the claims rest on, merged FLAT into `roofline` (the driver's record keeps
    scalars directly inside that object and drops nested ones): fractions of the HBM peak (fp32 MFMA peak for `fullscore_*`) by
    step time unless the key says otherwise, ms where the key ends in `_ms`.  Key names as VERDICT r5 "next" #1 lists them."""
    side = {}

    def put(key, *path):
        v = extra
        for k in path:
            if isinstance(k, int) and isinstance(v, (list, tuple)) and -len(v) <= k < len(v):
                v = v[k]
                continue
            if not isinstance(v, dict) or k not in v:
                return
            v = v[k]
        if isinstance(v, (int, float)):
            side[key] = v
    put('n1e8_uniform_frac', 'table_100M', 'n=64,B=65536', 'frac_of_hbm_peak')
    put('n1e8_walk_n1024_frac', 'table_100M', 'n=1024,B=4096', 'frac_of_hbm_peak')
    put('n1e8_popular_frac', 'table_100M', 'popular,n=64,B=65536', 'frac_of_hbm_peak')
    put('b4096_frac', 'sweep', 'B=4096', 'frac_of_hbm_peak')
    put('b16384_frac', 'sweep', 'B=16384', 'frac_of_hbm_peak')
    put('b4096_two_streams_frac', 'sweep', 'B=4096', 'two_streams_frac_of_hbm_peak')
    put('b16384_two_streams_frac', 'sweep', 'B=16384', 'two_streams_frac_of_hbm_peak')
    put('b4096_queue_frac', 'sweep', 'B=4096', 'queue', 'frac_of_hbm_peak')
    put('b16384_queue_frac', 'sweep', 'B=16384', 'queue', 'frac_of_hbm_peak')
    put('ssm_n256_frac', 'seq_softmax', 'frac_of_hbm_peak')
    put('train_step_frac', 'train_step', 'train_frac')
    put('sgd_step_frac', 'train_step', 'sgd_frac')
    put('sgd_step_prefetched_frac', 'train_step', 'sgd_prefetched_frac')
    put('sgd_step_prefetched_ms', 'train_step', 'sgd_step_prefetched_ms')
    put('adam_step_frac', 'train_step', 'adam_frac')
    put('shard_w1_frac', 'sharded_world1', 'frac_of_hbm_peak')
    put('shard_w1_train_frac', 'sharded_world1', 'train', 'frac_of_hbm_peak')
    put('shard_w1_train_ms', 'sharded_world1', 'train', 'ms_per_step')
    put('shard_w1_train_ssm_ms', 'sharded_world1', 'train_ssm', 'ms_per_step')
    put('shard_w1_train_ssm_frac', 'sharded_world1', 'train_ssm', 'frac_of_hbm_peak')
    put('fullscore_frac', 'fullscore', 'frac_of_peak')
    put('fullscore_tflops', 'fullscore', 'gemm_lse_tflops')
    put('fullscore_top100_ms', 'fullscore', 'with_top100_ms')
    put('fullscore_grad_items_tflops', 'fullscore', 'grad_items_tflops')
    put('fullscore_top100_frac', 'fullscore', 'with_top100_frac')
    put('fullscore_top100_only_ms', 'fullscore', 'top100_only_ms')
    put('fullscore_top100_only_frac', 'fullscore', 'top100_only_frac')
    put('softmax_train_step_ms', 'fullscore', 'softmax_train_step_ms')
    put('softmax_train_frac', 'fullscore', 'softmax_train_frac')
    put('softmax_train_step_peak_extra_mb', 'fullscore', 'softmax_train_step_peak_extra_MB')
    put('softmax_train_step_b8192_ms', 'fullscore', 'softmax_train_step_B8192_ms')
    put('softmax_train_b8192_frac', 'fullscore', 'softmax_train_B8192_frac')
    put('softmax_train_step_b8192_peak_extra_mb', 'fullscore', 'softmax_train_step_B8192_peak_extra_MB')
    put('softmax_train_step_recompute_ms', 'fullscore', 'softmax_train_step_recompute_ms')
    put('softmax_train_recompute_frac', 'fullscore', 'softmax_train_recompute_frac')
    put('softmax_train_step_store_peak_extra_mb', 'fullscore', 'softmax_train_step_store_peak_extra_MB')
    put('fullscore_flash_forward_tflops', 'fullscore', 'flash_forward_tflops')
    put('softmax_train_step_store_ms', 'fullscore', 'softmax_train_step_store_ms')
    put('softmax_train_store_frac', 'fullscore', 'softmax_train_store_frac')
    put('fullscore_grad_items_recompute_tflops', 'fullscore', 'grad_items_recompute_tflops')
    put('fullscore_grad_query_recompute_tflops', 'fullscore', 'grad_query_recompute_tflops')
    put('fit_c1_train_s', 'fit', 'c1_bpr_ml100k', 'train_s_per_epoch')
    put('fit_c1_valid_s', 'fit', 'c1_bpr_ml100k', 'valid_s_per_epoch')
    for b in (65536, 4096):
        put(f'fit_loop_B{b}_ms', 'fit', f'c2_B{b}', 'loop_ms_per_step')
        put(f'fit_stepper_B{b}_ms', 'fit', f'c2_B{b}', 'stepper_ms_per_step')
        put(f'fit_loop_over_stepper_b{b}', 'fit', f'c2_B{b}', 'loop_over_stepper')
        put(f'fit_loop_B{b}_M_triplets_s', 'fit', f'c2_B{b}', 'loop_M_triplets_s')
        put(f'fit_c2_b{b}_loss_first', 'fit', f'c2_B{b}', 'train_loss_first_last', 0)
        put(f'fit_c2_b{b}_loss_last', 'fit', f'c2_B{b}', 'train_loss_first_last', -1)
    put('fit_c1_loss_last', 'fit', 'c1_bpr_ml100k', 'train_loss_last')
    put('fit_c3_sasrec_ms_per_step', 'fit', 'c3_sasrec', 'ms_per_step')
    put('fit_c3_hot_path_ms', 'fit', 'c3_sasrec', 'hot_path_ms')
    put('fit_c3_hot_path_share_of_step', 'fit', 'c3_sasrec', 'hot_path_share_of_step')
    # (the split is taken on the WIDEST batch of the epoch, L = max_seq_len: its whole step next to its Transformer; the epoch's
    # mean step above is shorter because most batches are narrower)
    put('fit_c3_widest_batch_step_ms', 'fit', 'c3_sasrec', 'step_parts_ms', 'whole_step_one_batch')
    put('fit_c3_widest_batch_transformer_ms', 'fit', 'c3_sasrec', 'step_parts_ms', 'transformer_fwd_bwd_stock_torch')
    put('fit_c3_loss_first', 'fit', 'c3_sasrec', 'train_loss_first_last', 0)
    put('fit_c3_loss_last', 'fit', 'c3_sasrec', 'train_loss_first_last', -1)
    return side


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--items', type=int, default=None, help='default: 10 000 001 on one GPU, 100 000 001 sharded')
    ap.add_argument('--users', type=int, default=1_000_001)
    ap.add_argument('--dim', type=int, default=128)
    ap.add_argument('--batch', type=int, default=None, help='queries per step per GPU (default 65536 / 4096 sharded)')
    ap.add_argument('--neg', type=int, default=None, help='default 64 / 1024 sharded')
    ap.add_argument('--sampler', default=None, choices=['popular', 'uniform'], help='default popular / uniform sharded')
    ap.add_argument('--guide-log2', type=int, default=None, help='override the sampler guide-table size (experiments)')
    ap.add_argument('--pop-lookup', default='auto', choices=['auto', 'lines', 'lut', 'guide'],
                    help='inverse-CDF structure of the popularity sampler (experiments)')
    ap.add_argument('--shard-layout', default='block', choices=['block', 'interleaved'],
                    help='row ownership of the sharded table (--gpus N > 1): contiguous blocks, or rows r, r + N, ...')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-sweep', action='store_true')
    ap.add_argument('--no-fit', action='store_true', help='skip the BaseRetriever.fit figures (tools/bench_fit.py)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        self_launch(args.gpus)              # does not return
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)')
    sharded_run = world > 1
    if args.items is None:
        args.items = 100_000_001 if sharded_run else 10_000_001
    if args.batch is None:
        args.batch = 4096 if sharded_run else 65536
    if args.neg is None:
        args.neg = 1024 if sharded_run else 64
    if args.sampler is None:
        args.sampler = 'uniform' if sharded_run else 'popular'
    # RSA_BENCH_STAGED=1: the multi-rank branch on a single-GPU box -- all ranks share GPU 0 and their collectives are staged
    # over gloo (tools/staged_dist.py, the tests' harness).  Exercises the code path and the JSON keys; not a measurement.
    staged = world > 1 and os.environ.get('RSA_BENCH_STAGED') == '1'
    if staged:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if staged:
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            from staged_dist import StagedDist
            dist.init_process_group('gloo')
            dist = StagedDist(dist)
        else:
            dist.init_process_group('nccl', device_id=dev)
            assert dist.get_backend() == 'nccl'
        assert dist.get_world_size() == args.gpus

    import recstudio_amd as ra
    from recstudio_amd import _native as nat
    from recstudio_amd.retriever import _above_second_stream
    hi_cache = {}
    ra._native.lib()     # no extension, no benchmark
    if world > 1:        # torchrun pins OMP_NUM_THREADS=1: give each rank its share of the host cores for table builds
        torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))

    d, B, n = args.dim, args.batch, args.neg
    popular = args.sampler == 'popular'
    counts = zipf_counts(args.items, 100_000_000) if (popular or not sharded_run) else None
    gen = torch.Generator(device=dev).manual_seed(100 + rank)
    uid = torch.randint(1, args.users, (B,), device=dev, generator=gen)
    pos = torch.randint(1, args.items, (B,), device=dev, generator=gen)
    torch.manual_seed(2022 + rank)       # basemodel.yaml:63 seed

    extra = {}
    emit_lock, emitted = threading.Lock(), []

    def flush_c_stdio():
        # RCCL prints its banner through C stdio, which a pipe buffers until exit: push it out first so that the JSON line
        # is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass

    def headline_line(value, ms_step, workload, parallelism, roofline, extra):
        flush_c_stdio()
        line = {'metric': 'M scored (user,pos,neg) triplets/sec at d=128', 'value': round(value, 2),
                'unit': 'M triplets/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': round(ms_step, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                'dtype': 'f32', 'data': 'synthetic',
                'config': {'workload': workload, 'global_batch': B * world, 'num_neg': n, 'dim': d,
                           'n_items': args.items, 'parallelism': parallelism},
                'roofline': roofline}
        line.update(extra)
        return line
    force_shard = world == 1 and os.environ.get('RSA_BENCH_FORCE_SHARD') == '1'   # exercise the N>1 branch on one GPU
    if force_shard:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    if world == 1 and not force_shard:
        # FIRST (a fresh process, like a user's): the thing users run, BaseRetriever.fit end to end (tools/bench_fit.py) -- configs[0] on the committed ml-100k
        # fixture next to the reference's published epoch times, and configs[1]-shaped fit through loader + stepper
        if not args.no_sweep and not args.no_fit and args.dim == 128:
            try:
                torch.cuda.empty_cache()
                sys.path.insert(0, os.path.join(ROOT, 'tools'))
                from bench_fit import fit_figures
                seed_state = torch.cuda.get_rng_state(dev)
                extra['fit'] = fit_figures(ra, dev, n_items=args.items, n_users=args.users)
                torch.cuda.set_rng_state(seed_state, dev)
            except Exception as e:
                extra['fit'] = {'error': repr(e)[:300]}
        item, user = make_workload(dev, args.items, args.users, d)
        sampler = (ra.PopularSamplerModel(counts, guide_log2=args.guide_log2, lookup=args.pop_lookup) if popular
                   else ra.UniformSampler(args.items)).to(dev)
        kind = nat.SAMPLER_POPULAR if popular else nat.SAMPLER_UNIFORM
        kw = dict(query_index=uid, pos_ids=pos, sampler=kind)
        if popular:
            kw.update(sampler.lookup_kwargs())
        bufs = {}

        def fwd(b=B, u=uid, p=pos, key='main'):
            bufs[key] = ra.ops.fused_forward(item, user, n, out=bufs.get(key), **dict(kw, query_index=u, pos_ids=p))
            return bufs[key]

        def step():          # sample + gather + score + BPR loss (value and d loss/d score): one kernel + the mean
            bufs['step'] = ra.ops.fused_forward(item, user, n, out=bufs.get('step'), fused_bpr=True, **kw)
            return bufs['step']

        def step_unfused():  # the same with the loss as its own kernel
            o = fwd()
            return ra.ops.pairwise_loss(nat.LOSS_BPR, o['pos_score'], o['neg_score'], want_grad=True)

        # training step: forward + loss + row-sparse gradient scatter (+ user-row gradient)
        def train_two_pass():      # backward re-reads the negative rows for the user gradient
            o = step()
            return ra.ops.fused_backward(item, user, o['neg_ids'], o['dneg'], query_index=uid, pos_ids=pos,
                                         dpos=o['dpos'], dense_item_grad=False, row_item_grad=True, want_query_grad=True)

        def train():               # user gradient accumulated by the forward: every negative row is read once
            bufs['train'] = ra.ops.fused_forward(item, user, n, out=bufs.get('train'), fused_bpr=True,
                                                 want_query_grad=True, want_scores=False, **kw)
            o = bufs['train']
            return ra.ops.fused_backward(item, user, o['neg_ids'], o['dneg'], query_index=uid, pos_ids=pos,
                                         dpos=o['dpos'], dense_item_grad=False, row_item_grad=True, want_query_grad=False)
        try:
            ms_two = time_gpu(train_two_pass, max(10, args.steps // 4), 5) * 1e3
            ms_train = time_gpu(train, max(10, args.steps // 4), 5) * 1e3
            t_f = time_gpu(lambda: ra.ops.fused_forward(item, user, n, out=bufs['train'], fused_bpr=True,
                                                        want_query_grad=True, want_scores=False, **kw), max(10, args.steps // 4), 5) * 1e3
            prof_ms = lambda name: round((profile_record(name) or {}).get('step_kernel_us', 0.0) / 1e3, 4) or None      # noqa: E731
            extra['train_step'] = {'profile_step_kernel_ms': prof_ms('train_step_N1e7_popular_n64_B65536'),
                                   'profile_sgd_step_kernel_ms': prof_ms('sgd_step_N1e7_popular_n64_B65536'),
                                   'ms_per_step': round(ms_train, 4), 'value': round(B * n / ms_train / 1e3, 2),
                                   'unit': 'M triplets/s', 'what': 'forward + BPR loss + user-gradient rows (accumulated '
                                   'in the forward) + row-sparse item-gradient rows (no optimizer)',
                                   'forward_ms': round(t_f, 4), 'two_pass_ms_per_step': round(ms_two, 4)}
            # complete SGD step without gradient tensors (updates applied by the kernels); the tables are restored after
            from recstudio_amd.fused import bpr_sgd_step
            iw0, uw0 = item.clone(), user.clone()
            t_sgd = time_gpu(lambda: bpr_sgd_step(item, user, n, 1e-3, user_ids=uid, pos_ids=pos, sampler=sampler),
                             max(10, args.steps // 4), 3) * 1e3
            extra['train_step']['sgd_step_ms'] = round(t_sgd, 4)
            # the same steps with the weight-independent part of step k+1 (sampling, sort, solo classification) issued on a
            # side stream while step k runs: same results bit for bit, steady-state time per step
            from recstudio_amd.fused import PrefetchedBPRSGD
            stepper = PrefetchedBPRSGD(item, user, n, 1e-3, sampler)
            tick = {'t': stepper.prepare(uid, pos)}

            def prefetched_step():
                nxt = stepper.prepare(uid, pos)
                stepper.step(tick['t'])
                tick['t'] = nxt
            t_pf = time_gpu(prefetched_step, max(10, args.steps // 4), 3) * 1e3
            torch.cuda.synchronize()
            del stepper, tick
            extra['train_step']['sgd_step_prefetched_ms'] = round(t_pf, 4)
            extra['train_step']['sgd_step_prefetched_what'] = ('fused.PrefetchedBPRSGD: the sgd_step above with the next '
                                                               "step's negatives drawn, sorted and classified on a side stream "
                                                               'under the current step (wall clock per step; same weights bit for bit)')
            item.copy_(iw0)
            user.copy_(uw0)
            del iw0, uw0
            # complete lazy-Adam step (torch.optim.SparseAdam's rule) on the same machinery
            from recstudio_amd.fused import FusedBPRAdam
            iw0, uw0 = item.clone(), user.clone()
            fa = FusedBPRAdam(item, user, lr=1e-3)
            t_adam = time_gpu(lambda: fa.step(n, user_ids=uid, pos_ids=pos, sampler=sampler), max(10, args.steps // 4), 3) * 1e3
            extra['train_step']['adam_step_ms'] = round(t_adam, 4)
            atick = {'t': fa.prepare(n, user_ids=uid, pos_ids=pos, sampler=sampler)}

            def adam_ahead():
                nxt = fa.prepare(n, user_ids=uid, pos_ids=pos, sampler=sampler)
                fa.step_prepared(atick['t'])
                atick['t'] = nxt
            t_adam_pf = time_gpu(adam_ahead, max(10, args.steps // 4), 3) * 1e3
            torch.cuda.synchronize()
            extra['train_step']['adam_step_prefetched_ms'] = round(t_adam_pf, 4)
            item.copy_(iw0)
            user.copy_(uw0)
            del iw0, uw0, fa, atick
            torch.cuda.empty_cache()
            # Rooflines of the steps, SURVEY.md 8d's byte model: the forward's bytes + the read-modify-write of
            # the negative's row (2 * 4d) + the query-gradient and positive rows (2 * 4d / n) per triplet; lazy Adam touches
            # three rows (weight, exp_avg, exp_avg_sq) per updated row: 3 * 2 * 4d.  Whole steps against the HBM peak.
            fwd_b = bytes_per_triplet(d, n, popular, fused_loss=True)
            ts = extra['train_step']
            for key, per_triplet, t_ms, prof in (
                    ('train', fwd_b + 4 * d + 2 * 4 * d / n, ms_train, 'train_step_N1e7_popular_n64_B65536'),
                    ('sgd', fwd_b + 2 * 4 * d + 2 * 4 * d / n, t_sgd, 'sgd_step_N1e7_popular_n64_B65536'),
                    ('sgd_prefetched', fwd_b + 2 * 4 * d + 2 * 4 * d / n, t_pf, None),
                    ('adam', fwd_b + 3 * 2 * 4 * d + 3 * 2 * 4 * d / n, t_adam, 'adam_step_N1e7_popular_n64_B65536')):
                algb = per_triplet * B * n
                ts[f'{key}_alg_bytes_per_triplet'] = round(per_triplet, 1)
                ts[f'{key}_frac'] = round(algb / t_ms / 1e6 / HBM_PEAK_GBS, 4)
                rec = prof and profile_record(prof)
                if rec and rec.get('step_kernel_us'):
                    ts[f'{key}_profile_frac'] = round(algb / (rec['step_kernel_us'] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                    ts[f'{key}_profile_step_kernel_ms'] = round(rec['step_kernel_us'] / 1e3, 4)
                    ts[f'{key}_profile_source'] = f'{rec["_file"]}["{prof}"]'
            ts['frac_what'] = ('whole-step algorithmic bytes (SURVEY.md 8d: forward + 2*4d per triplet for the updated row + '
                               '2*4d/n for the query-gradient and positive rows; train: 4d, the row-sparse gradient written once; '
                               'adam: 3 rows read-modified-written) / step time / 8 TB/s')
            extra['train_step']['adam_step_what'] = ('forward + BPR loss + lazy Adam (SparseAdam rule) on the touched item '
                                                     'and user rows, gradient sums kept in registers (no gradient tensor)')
            extra['train_step']['sgd_step_what'] = ('negatives drawn and sorted by item id, forward + BPR loss with the SGD update of '
                                                    'every item row only one element touches applied by the wave that holds it, '
                                                    'sorted apply pass for the shared rows, sorted user-row update (no [N, d] '
                                                    'gradient tensor; bit-reproducible)')
        except Exception as e:      # never let the secondary figure kill the bench line
            extra['train_step'] = {'error': repr(e)[:200]}
        if not args.no_sweep:
            sweep = {}
            for b2 in (4096, 16384):
                u2, p2 = uid[:b2].contiguous(), pos[:b2].contiguous()
                # the same single-launch step as the headline line, through ops.FusedStep (the call frozen into its argument
                # block: at B = 4096 the ~35 us of Python per fused_forward call exceed the kernel's run time)
                st = ra.ops.FusedStep(item, user, n, fused_bpr=True, **dict(kw, query_index=u2, pos_ids=p2))
                t = time_gpu(st, args.steps, 10) * 1e3
                alg2 = bytes_per_triplet(d, n, popular, fused_loss=True) * b2 * n
                sweep[f'B={b2}'] = with_profile({'ms_per_step': round(t, 4), 'M_triplets_s': round(b2 * n / t / 1e3, 2),
                                                 'frac_of_hbm_peak': round(alg2 / t / 1e6 / HBM_PEAK_GBS, 4)},
                                                f'N1e7_popular_n64_B{b2}', alg2)
                # INDEPENDENT steps (forward-only scoring of many batches) alternating over two HIP streams: the sampling
                # chain of one launch (draw -> bucket line -> slot: ~7 us with the memory system idle) runs under the row
                # phase of the other instead of behind the stream's barrier packet.  Throughput mode only: the steps of a
                # training run depend on each other (there fused.PrefetchedBPRSGD overlaps the weight-independent part).
                try:
                    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
                    two = []
                    for s_ in streams:
                        with torch.cuda.stream(s_):
                            two.append(ra.ops.FusedStep(item, user, n, fused_bpr=True, **dict(kw, query_index=u2, pos_ids=p2)))
                    torch.cuda.synchronize()
                    cnt = {'i': 0}

                    def alternating():
                        i = cnt['i'] & 1
                        cnt['i'] += 1
                        with torch.cuda.stream(streams[i]):
                            two[i]()
                    t2s = time_gpu(alternating, args.steps, 10) * 1e3
                    sweep[f'B={b2}'].update(two_streams_ms_per_step=round(t2s, 4),
                                            two_streams_frac_of_hbm_peak=round(alg2 / t2s / 1e6 / HBM_PEAK_GBS, 4),
                                            two_streams_what='independent steps alternating over two streams (throughput mode)')
                    del two, streams
                except Exception as e:
                    sweep[f'B={b2}']['two_streams_error'] = repr(e)[:120]
                # a QUEUE of S independent batches consumed by ONE resident grid (rsa_fused_args.n_batches): every batch
                # draws its negatives from its own torch call (== S consecutive launches bit for bit, tests/test_gpu_round5.py),
                # the waves of different batches are out of phase, so one batch's sampling chain runs under another's row reads
                try:
                    S_q = max(2, 65536 // b2)
                    gq = torch.Generator(device=dev).manual_seed(77)
                    uq = torch.randint(1, args.users, (S_q * b2,), device=dev, generator=gq)
                    pq = torch.randint(1, args.items, (S_q * b2,), device=dev, generator=gq)
                    stq = ra.ops.FusedStep(item, user, n, fused_bpr=True, n_batches=S_q, **dict(kw, query_index=uq, pos_ids=pq))
                    tq = time_gpu(stq, max(5, args.steps // 2), 5) * 1e3 / S_q
                    sweep[f'B={b2}']['queue'] = with_profile(
                        {'batches': S_q, 'ms_per_batch': round(tq, 4), 'frac_of_hbm_peak': round(alg2 / tq / 1e6 / HBM_PEAK_GBS, 4),
                         'what': 'S independent batches consumed by one launch (throughput mode; ids == S consecutive launches)'},
                        f'queue_N1e7_popular_n64_B{b2}x{S_q}', alg2 * S_q)
                    del stq, uq, pq
                except Exception as e:
                    sweep[f'B={b2}']['queue_error'] = repr(e)[:160]
                del st
            extra['sweep'] = sweep
        # configs[4]: full-catalog scores on the fp32 MFMA (N = 1e6, d = 128), logsumexp fused, + exact top-100
        try:
            del bufs
            torch.cuda.empty_cache()
            n5, b5, k5 = 1_000_001, 2048, 100
            it5 = item[:n5]
            q5 = user[1:b5 + 1].contiguous()
            t_lse = time_gpu(lambda: ra.ops.fullscore(it5, q5, want_lse=True), 10, 3) * 1e3
            t_topk = time_gpu(lambda: ra.ops.fullscore(it5, q5, want_lse=True, k=k5), 5, 2) * 1e3
            # what BaseRetriever.topk issues at evaluation (baseretriever.py:374-397): the exact top-k alone, no logsumexp
            t_topk_only = time_gpu(lambda: ra.ops.fullscore(it5, q5, k=k5), 5, 2) * 1e3
            flops = 2.0 * b5 * d * (n5 - 1)
            extra['fullscore'] = with_profile(
                {'workload': f'B={b5} queries x N={n5} items, d={d}, fp32 MFMA (BASELINE.json configs[4])',
                 'gemm_lse_ms': round(t_lse, 3), 'gemm_lse_tflops': round(flops / t_lse / 1e9, 1),
                 'with_top100_ms': round(t_topk, 3), 'with_top100_frac': round(flops / t_topk / 1e9 / 157.3, 3),
                 'top100_only_ms': round(t_topk_only, 3), 'top100_only_frac': round(flops / t_topk_only / 1e9 / 157.3, 3),
                 'peak_tflops_fp32_matrix': 157.3,
                 'frac_of_peak': round(flops / t_lse / 1e9 / 157.3, 3)}, 'fullscore_lse_B2048_N1e6', 0, flops=flops)
        except Exception as e:
            extra['fullscore'] = {'error': repr(e)[:200]}
        # configs[2] shape: SASRec tail -- ragged history gather [B, L<=50, d] + sampled softmax, n = 256
        try:
            b3, L3, n3, n_it3 = 8192, 50, 256, 1_000_001
            g3 = torch.Generator(device=dev).manual_seed(3)
            lens = torch.randint(1, L3 + 1, (b3,), device=dev, generator=g3)
            end = torch.cumsum(lens, 0)
            start = end - lens
            flat = torch.randint(1, n_it3, (int(end[-1]),), device=dev, generator=g3)
            it3 = item[:n_it3]
            t_seg = time_gpu_best(lambda: ra.ops.seg_gather(it3, flat, start, end, L3), 50, 5) * 1e3
            seg_bytes = float(end[-1]) * (4 * d + 8) + b3 * L3 * 4 * d + b3 * L3 * 8
            q3 = user[1:b3 + 1].contiguous()
            pos3 = torch.randint(1, n_it3, (b3,), device=dev, generator=g3)
            ps3 = ra.PopularSamplerModel(counts[:n_it3]).to(dev)
            kw3 = dict(pos_ids=pos3, sampler=nat.SAMPLER_POPULAR, **ps3.lookup_kwargs())
            buf3 = {}

            def step3_unfused():
                buf3['o'] = ra.ops.fused_forward(it3, q3, n3, out=buf3.get('o'), **kw3)
                o = buf3['o']
                return ra.ops.pairwise_loss(nat.LOSS_SSM, o['pos_score'], o['neg_score'], o['pos_logp'], o['neg_logp'])

            def step3():          # ONE launch: sampling, gather, scores, logsumexp over the query's tiles, loss, d loss/d score
                buf3['f'] = ra.ops.fused_forward(it3, q3, n3, out=buf3.get('f'), fused_loss='ssm', **kw3)
                return buf3['f']

            def train3():         # + d loss/d query in the forward, write-only item-gradient rows in the backward
                buf3['t'] = ra.ops.fused_forward(it3, q3, n3, out=buf3.get('t'), fused_loss='ssm', want_query_grad=True, **kw3)
                o = buf3['t']
                return ra.ops.fused_backward(it3, q3, o['neg_ids'], o['dneg'], pos_ids=pos3, dpos=o['dpos'],
                                             dense_item_grad=False, row_item_grad=True, want_query_grad=False)

            def train3_two_pass():     # round-1 form: separate loss kernel, backward re-reads the rows for d loss/d query
                loss, dpos, dneg, _ = step3_unfused()
                o = buf3['o']
                return ra.ops.fused_backward(it3, q3, o['neg_ids'], dneg, pos_ids=pos3, dpos=dpos, dense_item_grad=False,
                                             row_item_grad=True, want_query_grad=True)
            t3 = time_gpu_best(step3, 50, 5) * 1e3
            t3u = time_gpu_best(step3_unfused, 50, 5) * 1e3
            t3t = time_gpu_best(train3, 30, 3) * 1e3
            t3t2 = time_gpu_best(train3_two_pass, 30, 3) * 1e3
            alg3 = bytes_per_triplet(d, n3, True) * b3 * n3
            extra['seq_softmax'] = with_profile({}, 'ssm_N1e6_popular_n256_B8192', alg3)
            extra['seq_softmax'].update({
                'workload': f'B={b3} prefixes, L<={L3}, N={n_it3}, d={d}, popularity sampler n={n3}, SampledSoftmax '
                            '(BASELINE.json configs[2] tail; the Transformer is stock PyTorch and not timed)',
                'seg_gather_ms': round(t_seg, 4), 'seg_gather_GBs': round(seg_bytes / t_seg / 1e6, 1),
                'sample_gather_score_ssm_ms': round(t3, 4), 'M_triplets_s': round(b3 * n3 / t3 / 1e3, 1),
                'alg_GBs': round(bytes_per_triplet(d, n3, True) * b3 * n3 / t3 / 1e6, 1),
                'frac_of_hbm_peak': round(bytes_per_triplet(d, n3, True) * b3 * n3 / t3 / 1e6 / HBM_PEAK_GBS, 4),
                'what': 'sampling + gather + scores + SampledSoftmax (logsumexp across the 4 tiles of a query, loss, '
                        'd loss/d score) in ONE launch',
                'unfused_loss_ms': round(t3u, 4),
                'train_step_ms': round(t3t, 4), 'train_step_two_pass_ms': round(t3t2, 4),
                'train_step_what': 'forward as above + d loss/d query accumulated in the forward + write-only '
                                   'row-sparse item-gradient rows (two_pass: separate loss kernel, backward re-reads the rows)'})
        except Exception as e:
            extra['seq_softmax'] = {'error': repr(e)[:200]}
        # full softmax training step on configs[4].  Default backward (round 6): NOTHING of [B, N] is ever written -- d/d query from
        # the query-stationary recompute pass, d/d items from the item-stationary one (five GEMMs of 2 B d N flop per step);
        # 'store' = the round-5 form (one [B, N-1] softmax write + read-back: four GEMMs, 8 GB at B = 2048)
        try:
            from recstudio_amd import scorer as sc_mod
            from recstudio_amd.scorer import full_lse
            w5 = item[:n5].detach().clone().requires_grad_(True)
            qq5 = q5.detach().clone().requires_grad_(True)

            def softmax_step(qq=qq5):
                w5.grad = qq.grad = None
                full_lse(qq, w5).mean().backward()
            fs = extra['fullscore']
            mode0 = sc_mod.FULL_SOFTMAX_BACKWARD

            def step_ms_and_peak(mode, qq=qq5, steps=5):
                sc_mod.FULL_SOFTMAX_BACKWARD = mode
                t = time_gpu(lambda: softmax_step(qq), steps, 2) * 1e3
                torch.cuda.reset_peak_memory_stats()
                base_mem = torch.cuda.memory_allocated()
                softmax_step(qq)
                torch.cuda.synchronize()
                return t, round((torch.cuda.max_memory_allocated() - base_mem) / 2 ** 20, 1)
            # 'flash' (the default): forward = logsumexp + d/d query in one pass, backward = one item-stationary recompute pass:
            # FOUR products of 2 B d N flop, no [B, N]
            t_sm, peak_sm = step_ms_and_peak('flash')
            fs['softmax_train_step_ms'] = round(t_sm, 3)
            fs['softmax_train_step_peak_extra_MB'] = peak_sm
            fs['softmax_train_frac'] = round(4 * flops / 157.3e12 * 1e3 / t_sm, 4)
            # B = 8192: 32 GB of [B, N] in the stored form; here the same two passes, 4 x the flops
            b8k = 8192
            q8k = user[1:b8k + 1].detach().clone().requires_grad_(True)
            t_sm8, peak8 = step_ms_and_peak('flash', q8k, 3)
            fs['softmax_train_step_B8192_ms'] = round(t_sm8, 3)
            fs['softmax_train_step_B8192_peak_extra_MB'] = peak8
            fs['softmax_train_B8192_frac'] = round(4 * flops * (b8k / b5) / 157.3e12 * 1e3 / t_sm8, 4)
            del q8k
            # 'recompute': lse-only forward + two recompute passes (five products, no [B, N]); 'store': round 5 (four products, one
            # [B, N-1] write + read-back: 8 GB at this shape)
            t_rc, _ = step_ms_and_peak('recompute')
            fs['softmax_train_step_recompute_ms'] = round(t_rc, 3)
            fs['softmax_train_recompute_frac'] = round(5 * flops / 157.3e12 * 1e3 / t_rc, 4)
            t_sm_st, peak_st = step_ms_and_peak('store')
            fs['softmax_train_step_store_ms'] = round(t_sm_st, 3)
            fs['softmax_train_step_store_peak_extra_MB'] = peak_st
            fs['softmax_train_store_frac'] = round(4 * flops / 157.3e12 * 1e3 / t_sm_st, 4)
            sc_mod.FULL_SOFTMAX_BACKWARD = mode0
            # where the step goes: all in-tree MFMA kernels
            lse5 = ra.ops.fullscore(w5.detach(), qq5.detach(), want_lse=True)[1]
            scale5 = torch.full((b5,), 1.0 / b5, device=dev)
            t_fl = time_gpu(lambda: ra.ops.fullscore_lse_grad(w5.detach(), qq5.detach()), 5, 2) * 1e3
            t_dq = time_gpu(lambda: ra.ops.fullscore_softmax(w5.detach(), qq5.detach(), lse5, scale5, want_query_grad=True,
                                                             want_probs=False), 5, 2) * 1e3
            gw5 = torch.empty_like(w5)
            t_dw = time_gpu(lambda: ra.ops.fullscore_softmax_dw(w5.detach(), qq5.detach(), lse5, scale5, out=gw5), 5, 2) * 1e3
            del gw5
            probs5 = ra.ops.fullscore_softmax(w5.detach(), qq5.detach(), lse5, scale5)
            t_rec_dq = time_gpu(lambda: ra.ops.fullscore_softmax(w5.detach(), qq5.detach(), lse5, scale5,
                                                                 want_query_grad=True), 5, 2) * 1e3
            gx5 = torch.empty(n5 - 1, d, device=dev)
            t_gx = time_gpu(lambda: ra.ops.probs_t_query(probs5, qq5.detach(), out=gx5), 5, 2) * 1e3
            del gx5, probs5
            fs['grad_items_tflops'] = round(flops / t_gx / 1e9, 1)
            fs['grad_items_recompute_tflops'] = round(2 * flops / t_dw / 1e9, 1)
            fs['grad_query_recompute_tflops'] = round(2 * flops / t_dq / 1e9, 1)
            fs['flash_forward_tflops'] = round(2 * flops / t_fl / 1e9, 1)
            fs['softmax_train_step_parts_ms'] = {
                'flash_forward__lse_and_grad_query': round(t_fl, 3),
                'forward_lse': round(t_lse, 3),
                'recompute__grad_query_no_write': round(t_dq, 3),
                'recompute__grad_items_no_probs': round(t_dw, 3),
                'store_form': {'recompute_write_and_grad_query': round(t_rec_dq, 3), 'grad_items_from_stored_probs': round(t_gx, 3)},
                'fp32_mfma_floor_ms': {'four_gemms': round(4 * flops / 157.3e12 * 1e3, 2), 'five_gemms': round(5 * flops / 157.3e12 * 1e3, 2)}}
            del w5, qq5
        except Exception as e:
            extra['fullscore']['softmax_train_step_error'] = repr(e)[:200]
        # ---- the headline line: exactly K steps after W warm-up steps, measured after the secondary figures so that the
        # GPU is at steady-state clocks (boxes of the pool differ by several per cent either way)
        hb = {}
        # a FRESH (user, positive) batch every step (one batch reused for every timed step keeps its
        # 67 MB of user / positive rows Infinity-Cache resident): drawn before the clock starts, one row per step -- a
        # row of the id matrix is a view, so the timed region holds exactly the kernels it held before
        n_batches = args.steps + args.warmup
        g_b = torch.Generator(device=dev).manual_seed(4242 + rank)
        uid_all = torch.randint(1, args.users, (n_batches, B), device=dev, generator=g_b)
        pos_all = torch.randint(1, args.items, (n_batches, B), device=dev, generator=g_b)
        tick = {'i': 0}

        def fresh_kw():
            i = tick['i'] % n_batches
            tick['i'] += 1
            return dict(kw, query_index=uid_all[i], pos_ids=pos_all[i])

        def step():          # sample + gather + score + BPR loss (value, d loss/d score, mean): ONE kernel
            hb['step'] = ra.ops.fused_forward(item, user, n, out=hb.get('step'), fused_bpr=True, **fresh_kw())
            return hb['step']

        def step_unfused():  # the same with the loss as its own kernel
            hb['fwd'] = ra.ops.fused_forward(item, user, n, out=hb.get('fwd'), **fresh_kw())
            o = hb['fwd']
            return ra.ops.pairwise_loss(nat.LOSS_BPR, o['pos_score'], o['neg_score'], want_grad=True)

        tick['i'] = 0
        ms_step = time_gpu(step, args.steps, args.warmup) * 1e3
        extra['unfused_loss_ms_per_step'] = round(time_gpu(step_unfused, args.steps, args.warmup) * 1e3, 4)

        # dominant kernel alone, timed with events on the launch stream
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        torch.cuda.synchronize()
        for a, b in evs:      # exactly one launch between the events: the fused kernel (no mean reduction)
            a.record()
            hb['step'] = ra.ops.fused_forward(item, user, n, out=hb['step'], fused_bpr=True, want_mean=False, **fresh_kw())
            b.record()
        torch.cuda.synchronize()
        k_ms = sorted(a.elapsed_time(b) for a, b in evs)
        k_avg = sum(k_ms) / len(k_ms)
        alg = bytes_per_triplet(d, n, popular, fused_loss=True) * B * n
        achieved = alg / (k_avg * 1e-3) / 1e9
        roofline = {'bound': 'hbm', 'kernel': 'rsa::fused_fwd_kernel<32,false,false,true,true> (sample+gather+score+BPR epilogue)',
                    'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': None,
                    'alg_bytes_per_launch': int(alg), 'avg_kernel_ms': round(k_avg, 4),
                    'median_kernel_ms': round(k_ms[len(k_ms) // 2], 4)}
        # The same launch on OTHER allocations of its output buffers (on some boxes certain allocations are 10-15 % slower
        # to write into under read load -- bimodal, stable for the life of the allocation).  The headline ran on a PLAIN torch
        # allocation (recstudio_amd.placement is opt-in since round 6); three more plain ones, all kept alive so that each is a
        # new allocation, then one chosen by placement -- flat scalars, so that the driver's record keeps them:
        # placement goes back on by default only if placed_kernel_ms <= 0.97 * plain_kernel_ms_min in that record.
        try:
            from recstudio_amd import placement

            def kernel_ms_on(o2):
                def on_o2():
                    ra.ops.fused_forward(item, user, n, out=o2, fused_bpr=True, want_mean=False, **fresh_kw())
                prewarm(on_o2, 0.3)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.steps):
                    on_o2()
                e1.record()
                torch.cuda.synchronize()
                return round(e0.elapsed_time(e1) / args.steps, 4)
            held, plain_ms = [hb['step']], [round(k_avg, 4)]
            for _ in range(3):
                with placement.disabled():
                    held.append(ra.ops.fused_forward(item, user, n, fused_bpr=True, want_mean=False, **fresh_kw()))
                plain_ms.append(kernel_ms_on(held[-1]))
            roofline['plain_kernel_ms_min'], roofline['plain_kernel_ms_max'] = min(plain_ms), max(plain_ms)
            with placement.enabled():
                held.append(ra.ops.fused_forward(item, user, n, fused_bpr=True, want_mean=False, **fresh_kw()))
            roofline['placed_kernel_ms'] = kernel_ms_on(held[-1])
            summ = placement.summary(dev)
            roofline['placement_probes'], roofline['placement_rejected_slow'] = summ['probes'], summ['rejected_slow']
            roofline['placement_default'] = 'off (RSA_PLACEMENT=1 to enable)' if not placement.ENABLED else 'on (RSA_PLACEMENT=1)'
            del held
            placement.release(dev)
            torch.cuda.empty_cache()
        except Exception as e:
            roofline['placement_error'] = repr(e)[:120]
        # The tracked profile of this very kernel and shape (profiles/<round>_kernel_profiles.json: rocprofv3 --kernel-trace --stats
        # average and the FETCH_SIZE / WRITE_SIZE passes, collected by tools/collect_shapes.sh on another box of the
        # pool): `profile_frac` is the same algorithmic bytes over THAT average, printed next to the live `frac`; the PMC
        # traffic is attached only while the two kernel times are within 15 % (a changed kernel must be re-profiled)
        if (args.items, args.batch, args.neg, args.dim, popular, args.pop_lookup) == (10_000_001, 65536, 64, 128, True, 'auto'):
            rec = profile_record('headline_N1e7_popular_n64_B65536')
            if rec and rec.get('avg_us'):
                roofline['profile_frac'] = round(alg / (rec['avg_us'] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                roofline['profile_avg_kernel_ms'] = round(rec['avg_us'] / 1e3, 4)
                roofline['profile_source'] = f'{rec["_file"]}["headline_N1e7_popular_n64_B65536"] (rocprofv3 --kernel-trace --stats)'
                if rec.get('hbm_bytes_per_launch') and abs(rec['avg_us'] / 1e3 - k_avg) / k_avg < 0.15:
                    roofline['traffic'] = int(rec['hbm_bytes_per_launch'])
                    roofline['traffic_source'] = ('same file: FETCH_SIZE + WRITE_SIZE passes, units and gfx950 correction as in '
                                                  'MI355X_MICROARCH.md (see tools/collect_shapes.sh)')

        # north_star target: the fused gather+sample+score(+BPR) on a 100 M-item table (51.2 GB) at d = 128
        if not args.no_sweep and args.dim == 128:
            try:
                del item
                torch.cuda.empty_cache()
                n8 = 100_000_001
                item8 = torch.empty(n8, d, device=dev).normal_(0, 0.02, generator=torch.Generator(device=dev).manual_seed(8))
                item8[0] = 0
                pos8 = torch.randint(1, n8, (B,), device=dev, generator=gen)
                res8 = {}
                for name, n_neg, b_q in (('n=64,B=65536', 64, B), ('n=1024,B=4096', 1024, 4096)):
                    u8, p8 = uid[:b_q].contiguous(), pos8[:b_q].contiguous()
                    b8 = {}

                    def st8(n_neg=n_neg, u8=u8, p8=p8, b8=b8):
                        b8['o'] = ra.ops.fused_forward(item8, user, n_neg, out=b8.get('o'), fused_bpr=True, want_mean=False,
                                                       query_index=u8, pos_ids=p8, sampler=nat.SAMPLER_UNIFORM)
                    t8 = time_gpu_best(st8, 50, 5) * 1e3
                    alg8 = bytes_per_triplet(d, n_neg, False, fused_loss=True) * b_q * n_neg
                    res8[name] = with_profile({'ms': round(t8, 4), 'M_triplets_s': round(b_q * n_neg / t8 / 1e3, 1),
                                               'alg_GBs': round(alg8 / t8 / 1e6, 1),
                                               'frac_of_hbm_peak': round(alg8 / t8 / 1e6 / HBM_PEAK_GBS, 4)},
                                              f'N1e8_uniform_n{n_neg}_B{b_q}', alg8)
                try:          # the popularity sampler on the same table (2^25-bucket lookup table, 537 MB)
                    ps8 = ra.PopularSamplerModel(zipf_counts(n8, 100_000_000), lookup=args.pop_lookup).to(dev)
                    b8p = {}

                    def st8p():
                        b8p['o'] = ra.ops.fused_forward(item8, user, n, out=b8p.get('o'), fused_bpr=True, want_mean=False,
                                                       query_index=uid, pos_ids=pos8, sampler=nat.SAMPLER_POPULAR,
                                                       **ps8.lookup_kwargs())
                    t8p = time_gpu_best(st8p, 50, 5) * 1e3
                    alg8p = bytes_per_triplet(d, n, True, fused_loss=True) * B * n
                    res8['popular,n=64,B=65536'] = with_profile(
                        {'ms': round(t8p, 4), 'M_triplets_s': round(B * n / t8p / 1e3, 1), 'alg_GBs': round(alg8p / t8p / 1e6, 1),
                         'frac_of_hbm_peak': round(alg8p / t8p / 1e6 / HBM_PEAK_GBS, 4),
                         'lookup': 'bucket lines 2^%d x 128 B' % ps8.lines_log2 if ps8.lines_log2 else 'lut 2^%d' % ps8.guide_log2},
                        'N1e8_popular_n64_B65536', alg8p)
                    del ps8, b8p
                except Exception as e:
                    res8['popular,n=64,B=65536'] = {'error': repr(e)[:200]}
                extra['table_100M'] = {'workload': f'uniform (and popularity) sampler + gather + score + fused BPR, N={n8} items (51.2 GB table), d={d} '
                                       '(north_star target; BASELINE.json configs[3] per-GPU shape for n=1024)',
                                       'timing': 'best of 3 windows of 50 steps each (side figure; the headline is one K-step window)',
                                       **res8}
                del item8
            except Exception as e:
                extra['table_100M'] = {'error': repr(e)[:200]}
        # the multi-GPU workload (configs[3]) on ONE rank: a 12.5 M-row block (1/8 of the 100 M-item table), n = 1024,
        # B = 4096 through the sharded step -- the N = 1 reference point of the `--gpus N` line (same code path, same per-GPU
        # shape).  With one rank every collective of the step is the identity and is skipped; `with_collectives_ms` forces
        # them through a world-size-1 RCCL group (what the step's own all-gather / all-to-all launches cost without a wire)
        if not args.no_sweep and args.dim == 128:
            try:
                import torch.distributed as dist1
                from recstudio_amd import shard
                os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
                os.environ.setdefault('MASTER_PORT', '29541')
                dist1.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
                dist = dist1
                n_blk, n1k, b1k = 12_500_001, 1024, 4096
                blk = torch.empty(n_blk, d, device=dev).normal_(0, 0.02, generator=torch.Generator(device=dev).manual_seed(9))
                blk[0] = 0
                us = ra.UniformSampler(n_blk)
                u1, p1 = uid[:b1k].contiguous(), torch.randint(1, n_blk, (b1k,), device=dev, generator=gen)
                res1 = {}
                for key, force in (('ms_per_step', False), ('with_collectives_ms', True)):
                    tbl = shard.ShardedItemTable(blk, shard.RowShardPlan(n_blk, 1), 0, dist1, check_every=0, force_collectives=force)

                    def st1(tbl=tbl):     # sample + route + (exchange) + owner-side score + (exchange) + gather home + BPR loss + gradient
                        return tbl.sample_and_score(user, u1, p1, n1k, us, fused_loss='bpr', want_ids=False, want_grad=True)
                    st1()
                    res1[key] = time_gpu(st1, 50, 5) * 1e3
                    tbl.check_overflow()
                    if key == 'ms_per_step':
                        # one batch ahead: the next step's draw + routing (+ key exchange) on a second stream under this one
                        look1 = {'t': tbl.prepare_forward(p1, n1k, us, fused_loss='bpr', want_ids=False)}

                        def st1_ahead(tbl=tbl, look1=look1):
                            nxt = tbl.prepare_forward(p1, n1k, us, fused_loss='bpr', want_ids=False)
                            tbl.sample_and_score(user, u1, p1, n1k, us, fused_loss='bpr', want_ids=False, want_grad=True, ticket=look1['t'])
                            look1['t'] = nxt
                        with _above_second_stream(True, dev, hi_cache):
                            st1_ahead()
                            res1['one_batch_ahead_ms'] = time_gpu(st1_ahead, 50, 5) * 1e3
                            tbl.sample_and_score(user, u1, p1, n1k, us, fused_loss='bpr', want_ids=False, want_grad=True, ticket=look1['t'])
                        tbl.check_overflow()
                t1 = res1['ms_per_step']
                alg1 = bytes_per_triplet(d, n1k, False) * b1k * n1k
                extra['sharded_world1'] = with_profile(
                    {'workload': f'configs[3] per-GPU shape on one rank: {n_blk}-row block, neg={n1k}, B={b1k}, uniform sampler drawn in '
                                 'the routing launch, fixed-capacity exchange v2, fused home kernel (scores, BPR loss, d loss/d score)',
                     'ms_per_step': round(t1, 4), 'M_triplets_s': round(b1k * n1k / t1 / 1e3, 2),
                     'frac_of_hbm_peak': round(alg1 / t1 / 1e6 / HBM_PEAK_GBS, 4),
                     'with_collectives_ms': round(res1['with_collectives_ms'], 4),
                     'one_batch_ahead_ms': round(res1['one_batch_ahead_ms'], 4),
                     'kernels_per_step': 'embedding_gather, shard_sample_route, fused_fwd_kernel (segment form), shard_home'},
                    'sharded_world1_step', alg1, whole_step=True)
                # ... and the complete in-place-SGD TRAINING step of that shape (the per-GPU ceiling of an 8-GPU training
                # curve): the stock BPR step evaluated on the owner of the negatives -- sampling + routing, sorts by row and
                # by query, ONE pass over the item rows (scores, loss, query-gradient partials, rows one element touches
                # updated in place), sorted apply for the shared rows, user rows.  SURVEY.md 8d bytes at n = 1024.
                tower1 = torch.nn.Embedding(args.users, d).to(dev)
                with torch.no_grad():
                    tower1.weight.copy_(user)
                blk0 = blk.clone()
                tbl_t = shard.ShardedItemTable(blk, shard.RowShardPlan(n_blk, 1), 0, dist1, check_every=0)
                trainer1 = shard.ShardedRetriever(tbl_t, tower1, us, ra.BPRLoss(), n1k, item_sgd_lr=1e-3, query_sgd_lr=1e-3)
                trainer1.training_step(u1, p1)
                t_tr = time_gpu(lambda: trainer1.training_step(u1, p1), 30, 3) * 1e3
                tbl_t.check_overflow()
                # ... and one batch ahead: the next step's negatives, routing, key exchange and owner-side sorts on a second
                # stream under the current step (ShardedRetriever.prepare_step; wall clock per step, the same weights)
                look = {'t': trainer1.prepare_step(u1, p1)}

                def ahead_step():
                    nxt = trainer1.prepare_step(u1, p1)
                    trainer1.training_step(u1, p1, ticket=look['t'])
                    look['t'] = nxt
                with _above_second_stream(True, dev, hi_cache):     # the steps on a high-priority stream, the look-ahead below it
                    t_la = time_gpu(ahead_step, 30, 3) * 1e3
                    trainer1.training_step(u1, p1, ticket=look['t'])
                tbl_t.check_overflow()
                blk.copy_(blk0)
                # ... and the same step with the stock SampledSoftmaxLoss, on the owners (two phases around the (max, sum)
                # all-reduce: one walk over the rows, then the sorted apply pass) and with the scores sent home (home kernel,
                # one-call backward) -- VERDICT r4 "next" #3
                ssm_ms = {}
                for own in (True, False):
                    tbl_s = shard.ShardedItemTable(blk, shard.RowShardPlan(n_blk, 1), 0, dist1, check_every=0)
                    tr_s = shard.ShardedRetriever(tbl_s, tower1, us, ra.SampledSoftmaxLoss(), n1k, item_sgd_lr=1e-3, query_sgd_lr=1e-3,
                                                  owner_ssm=own)
                    tr_s.training_step(u1, p1)
                    ssm_ms[own] = time_gpu(lambda: tr_s.training_step(u1, p1), 20, 3) * 1e3
                    tbl_s.check_overflow()
                    blk.copy_(blk0)
                    del tbl_s, tr_s
                del blk0, tower1, trainer1
                alg_tr = (bytes_per_triplet(d, n1k, False) + 2 * 4 * d + 2 * 4 * d / n1k) * b1k * n1k
                tr = with_profile({'ms_per_step': round(t_tr, 4), 'M_triplets_s': round(b1k * n1k / t_tr / 1e3, 2),
                                   'frac_of_hbm_peak': round(alg_tr / t_tr / 1e6 / HBM_PEAK_GBS, 4),
                                   'alg_bytes_per_triplet': round(alg_tr / b1k / n1k, 1),
                                   'one_batch_ahead_ms': round(t_la, 4),
                                   'one_batch_ahead_frac': round(alg_tr / t_la / 1e6 / HBM_PEAK_GBS, 4),
                                   'what': 'in-place SGD training step of the same shape, BPR evaluated on the owners '
                                           '(shard.ShardedItemTable.bpr_step_on_owners): item rows read once per step; '
                                           'one_batch_ahead: the weight-independent half of the next step (negatives, routing, '
                                           'key exchange, owner-side sorts) on a second stream under the current one (the steps on a high-priority stream)'},
                                  'sharded_world1_train', alg_tr, whole_step=True)
                extra['sharded_world1']['train'] = tr
                extra['sharded_world1']['train_ssm'] = with_profile({
                    'ms_per_step': round(ssm_ms[True], 4), 'frac_of_hbm_peak': round(alg_tr / ssm_ms[True] / 1e6 / HBM_PEAK_GBS, 4),
                    'scores_at_home_ms_per_step': round(ssm_ms[False], 4),
                    'what': 'the same in-place SGD step with SampledSoftmaxLoss evaluated on the owners (ssm_step_on_owners: rows '
                            'read once for scores + query gradient; solo rows read-modified-written by a second walk by query, '
                            'shared rows by the sorted apply pass) vs the '
                            'score-at-home protocol; same SURVEY 8d bytes as the BPR step'},
                    'sharded_world1_train_ssm', alg_tr, whole_step=True)
                del blk, tbl, tbl_t
            except Exception as e:
                extra['sharded_world1'] = {'error': repr(e)[:200]}
        value = B * n / ms_step / 1e3
        parallelism = 'single'
        workload = (f'BPR two-tower d={d}, synthetic {args.items} items / {args.users} users / 1e8-interaction Zipf '
                    f'popularity, {args.sampler} sampler neg={n}, InnerProduct + BPR loss, B={B} queries/step '
                    f'(BASELINE.json configs[1])')
        # box state under load + the side figures the claims rest on, numbers only, INSIDE `roofline` (the driver's record
        # keeps that object whole and only the names of the other extra keys)
        try:
            item_b = torch.empty(2_000_001, d, device=dev).normal_(0, 0.02)
            pb, bb = pos % 2_000_000 + 1, {}

            def busy():      # any HBM-bound launch of the path: what the clocks are under THIS kind of load
                bb['o'] = ra.ops.fused_forward(item_b, user, n, out=bb.get('o'), query_index=uid, pos_ids=pb,
                                               sampler=nat.SAMPLER_UNIFORM, fused_bpr=True, want_mean=False)
            box = box_state(busy, device=local)
            del item_b, bb
        except Exception as e:
            box = {'error': repr(e)[:120]}
        # FLAT: the driver's record keeps the scalars directly inside `roofline` and drops every nested object
        roofline.update({k: v for k, v in box.items() if k not in roofline})
        roofline.update({k: v for k, v in side_figures(extra).items() if k not in roofline})
        if rank == 0 and not args.no_cpu_baseline:
            extra['cpu_baseline'] = cpu_baseline(args, counts, d, B, n)
    else:
        # BASELINE.json configs[3]: the item table row-sharded over the ranks, ids out / scores back by RCCL all-to-all.
        # How to read the line: `value` = world * B * n / ms_per_step of the SINGLE-slice step (K steps, barrier-bracketed,
        # MAX over ranks); `world1_reference` = the identical per-GPU shape (same row block, same B, same n) timed IN THIS
        # JOB on every rank alone (a one-rank table: no collectives), MAX over ranks; `efficiency_vs_world1` = that time /
        # ms_per_step, i.e. the weak-scaling efficiency of the exchange itself, independent of the configs[1] N = 1 line.
        from recstudio_amd import shard
        plan = shard.RowShardPlan(args.items, world, layout=args.shard_layout)
        n_loc = plan.n_local(rank)
        g = torch.Generator(device=dev).manual_seed(7 + rank)
        item_local = torch.empty(n_loc, d, device=dev).normal_(0, 0.02, generator=g)
        if rank == 0:
            item_local[0] = 0
        user = torch.empty(args.users, d, device=dev).normal_(0, 0.02, generator=torch.Generator(device=dev).manual_seed(3))
        sampler = (ra.PopularSamplerModel(counts, lookup=args.pop_lookup) if popular else ra.UniformSampler(args.items)).to(dev)

        def make_step(tbl, smp, u, p, nn):
            def step():      # sample + route, key exchange, owner-side score, score exchange, gather home + BPR loss + gradient
                return tbl.sample_and_score(user, u, p, nn, smp, fused_loss='bpr', want_ids=False, want_grad=True)
            return step

        def max_over_ranks(ms):
            t = torch.tensor([ms], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        # ---- the world-1 reference, in this job: every rank alone on its own block (rows re-based to a one-rank plan)
        solo_plan = shard.RowShardPlan(n_loc, 1)
        solo_pos = (pos % (n_loc - 1)) + 1
        solo_sampler = (ra.PopularSamplerModel(plan.take(counts, rank), lookup=args.pop_lookup) if popular else ra.UniformSampler(n_loc)).to(dev)
        solo_tbl = shard.ShardedItemTable(item_local, solo_plan, 0, dist, check_every=0)
        solo_step = make_step(solo_tbl, solo_sampler, uid, solo_pos, n)
        solo_step()
        ms_solo = max_over_ranks(time_gpu(solo_step, args.steps, args.warmup, dist) * 1e3)
        solo_tbl.check_overflow()
        del solo_tbl, solo_sampler

        def measure(chunks):
            """K timed steps (barrier + synchronize on both sides, MAX over ranks) of the sharded step with the step's
            elements cut into `chunks` pipelined slices; checked for dropped elements (the count is job-wide on every
            rank), retried with more slack."""
            for slack in (1.08, 1.3, 2.0):
                # check_every=0: no overflow check inside the timed region; the check below covers all of it
                tbl = shard.ShardedItemTable(item_local, plan, rank, dist, slack=slack, check_every=0, chunks=chunks,
                                             force_collectives=True)
                step = make_step(tbl, sampler, uid, pos, n)
                step()                   # calibration launch of the fixed-capacity exchange (exact counts, once) + first step
                ms = time_gpu(step, args.steps, args.warmup, dist) * 1e3
                try:
                    tbl.check_overflow() # nothing was dropped on any rank during the timed steps (same value everywhere)
                    break
                except RuntimeError:     # raised on every rank alike: time again with more slack
                    continue
            return max_over_ranks(ms), tbl

        ms_step, table = measure(1)
        alg = bytes_per_triplet(d, n, popular) * B * n
        achieved = alg / (ms_step * 1e-3) / 1e9
        value = world * B * n / ms_step / 1e3
        roofline = {'bound': 'hbm', 'kernel': 'whole sharded step (per GPU): embedding_gather + all-gather, shard_sample_route, key '
                    'all-to-all, fused_fwd_kernel on the received segments, score all-to-all, shard_home (scores + BPR + gradient)',
                    'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                    'traffic': None}
        try:
            rccl = '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            rccl = 'unknown'
        extra.update({
            'per_gpu_M_triplets_s': round(B * n / ms_step / 1e3, 2),
            'world1_reference': {'ms_per_step': round(ms_solo, 4), 'per_gpu_M_triplets_s': round(B * n / ms_solo / 1e3, 2),
                                 'what': 'the same per-GPU shape (own row block, same B and n, same step) on every rank ALONE, '
                                         'timed in this job before the N-rank measurement, MAX over ranks'},
            'efficiency_vs_world1': round(ms_solo / ms_step, 4),
            'ranks_seen': int(dist.get_world_size()), 'backend': 'staged-gloo (test harness)' if staged else 'nccl (RCCL)',
            'rccl_version': rccl,
            'exchange': {'mode': table.exchange, 'layout': plan.layout, 'slack': table.slack, 'capacity_per_segment': table._cap.get((B, n, 1)),
                         'mean_per_owner': B * (n + 1) // world,
                         'what': 'equal-split all-to-all of fixed-capacity, self-describing segments ({live, dropped} header + '
                                 '8-byte keys); no split sizes on the host (one calibration launch before the timed region)'}})
        parallelism = f'item-table row-sharded x{world} + RCCL all-to-all (ids out, scores back)'
        workload = (f'two-tower d={d}, {args.items}-item table row-sharded over {world} GPUs, {args.sampler} sampler '
                    f'neg={n}, InnerProduct + BPR loss, B={B} queries/step/GPU (BASELINE.json configs[3])')

        def emit():
            with emit_lock:
                if emitted:
                    return
                emitted.append(1)
                if rank == 0:
                    # FLAT scalars inside `roofline` (what the driver's record keeps): the in-job world-1 reference and the side figures
                    def put(key, *path):
                        v = extra
                        for k in path:
                            if not isinstance(v, dict) or k not in v:
                                return
                            v = v[k]
                        if isinstance(v, (int, float)) and not isinstance(v, bool) and key not in roofline:
                            roofline[key] = v
                    put('per_gpu_M_triplets_s', 'per_gpu_M_triplets_s')
                    put('world1_ms_per_step', 'world1_reference', 'ms_per_step')
                    put('efficiency_vs_world1', 'efficiency_vs_world1')
                    put('one_batch_ahead_ms', 'one_batch_ahead', 'ms_per_step')
                    put('train_step_ms', 'train_step', 'ms_per_step')
                    put('train_step_world1_ms', 'train_step', 'world1_reference_ms')
                    put('train_step_efficiency_vs_world1', 'train_step', 'efficiency_vs_world1')
                    put('train_step_frac_per_gpu', 'train_step', 'frac_of_hbm_peak_per_gpu')
                    put('train_step_one_batch_ahead_ms', 'train_step', 'one_batch_ahead_ms')
                    put('train_step_ssm_ms', 'train_step_ssm', 'ms_per_step')
                    put('train_step_ssm_world1_ms', 'train_step_ssm', 'world1_reference_ms')
                    put('train_step_ssm_efficiency_vs_world1', 'train_step_ssm', 'efficiency_vs_world1')
                    put('exact_exchange_ms_per_step', 'exact_exchange_ms_per_step')
                    put('sharded_n64_ms_per_step', 'sharded_n64', 'ms_per_step')
                    put('other_sampler_ms_per_step', 'other_sampler', 'ms_per_step')
                    print(json.dumps(headline_line(value, ms_step, workload, parallelism, roofline, extra)), flush=True)
        # The side figures below run collectives: should one of them stall (a rank that failed alone leaves the others
        # waiting), the measured headline must still come out -- a watchdog thread prints the line and ends the rank.

        def give_up():
            extra['extras_timed_out'] = True
            emit()
            os._exit(0)
        watchdog = threading.Timer(float(os.environ.get('RSA_BENCH_EXTRAS_DEADLINE_S', '420')), give_up)
        watchdog.daemon = True
        watchdog.start()

        def timed_max(fn, steps, warm):
            fn()
            return max_over_ranks(time_gpu(fn, steps, warm, dist) * 1e3)
        # The same K steps ONE BATCH AHEAD: step t + 1's draw, routing and key exchange -- the largest message of the step --
        # issued on a second stream before step t is scored (ShardedItemTable.prepare_forward).  Beside the headline.
        try:
            tbl_a = shard.ShardedItemTable(item_local, plan, rank, dist, slack=table.slack, check_every=0, force_collectives=True)
            look_a = {'t': tbl_a.prepare_forward(pos, n, sampler, fused_loss='bpr', want_ids=False)}

            def ahead_fwd():
                nxt = tbl_a.prepare_forward(pos, n, sampler, fused_loss='bpr', want_ids=False)
                tbl_a.sample_and_score(user, uid, pos, n, sampler, fused_loss='bpr', want_ids=False, want_grad=True, ticket=look_a['t'])
                look_a['t'] = nxt
            with _above_second_stream(True, dev, hi_cache):
                ms_a = timed_max(ahead_fwd, args.steps, args.warmup)
                tbl_a.sample_and_score(user, uid, pos, n, sampler, fused_loss='bpr', want_ids=False, want_grad=True, ticket=look_a['t'])
            tbl_a.check_overflow()
            extra['one_batch_ahead'] = {'ms_per_step': round(ms_a, 4), 'M_triplets_s': round(world * B * n / ms_a / 1e3, 2),
                                        'efficiency_vs_world1': round(ms_solo / ms_a, 4),
                                        'what': 'side figure: the same K steps with the next step\'s draw, routing and key exchange on '
                                                'a second stream under the current step; the headline is the in-place step'}
            del tbl_a
        except Exception as e:
            extra['one_batch_ahead'] = {'error': repr(e)[:200]}
        # The same K steps with the step cut into 2 / 4 slices whose exchanges overlap the scoring of the neighbouring
        # slices (ShardedItemTable(chunks=...)): reported BESIDE the headline, which stays the single-slice time
        try:
            tried = {'1': round(ms_step, 4)}
            for c in (2, 4):
                ms_c, _ = measure(c)
                tried[str(c)] = round(ms_c, 4)
            extra['pipelined_slices'] = {'ms_per_step_by_slices': tried,
                                         'what': 'side figure: the headline is the 1-slice time whatever these say'}
        except Exception as e:
            extra['pipelined_slices'] = {'error': repr(e)[:200]}
        # the TRAINING step on the sharded table: in-place SGD, the stock BPR step evaluated on the owners of the negatives
        # (shard.ShardedItemTable.bpr_step_on_owners: 8 bytes per triplet over xGMI, the item rows read once per step), with
        # its own in-job world-1 reference (every rank alone on its block)
        try:
            w_keep = item_local.clone()
            tower_t = torch.nn.Embedding(args.users, d).to(dev)
            with torch.no_grad():
                tower_t.weight.copy_(user)
            res_t = {}
            for tag, pl, rk, smp_t, pos_t, force in (('world1', shard.RowShardPlan(n_loc, 1), 0,
                                                       (ra.PopularSamplerModel(plan.take(counts, rank), lookup=args.pop_lookup).to(dev)
                                                        if popular else ra.UniformSampler(n_loc)), solo_pos, False),
                                                      ('sharded', plan, rank, sampler, pos, True)):
                tbl_t = shard.ShardedItemTable(item_local, pl, rk, dist, check_every=0, force_collectives=force)
                trn_t = shard.ShardedRetriever(tbl_t, tower_t, smp_t, ra.BPRLoss(), n, item_sgd_lr=1e-3, query_sgd_lr=1e-3)
                res_t[tag] = timed_max(lambda trn_t=trn_t, pos_t=pos_t: trn_t.training_step(uid, pos_t), max(10, args.steps // 4), 3)
                tbl_t.check_overflow()
                if tag == 'sharded' and trn_t.can_prepare():
                    # one batch ahead: the key exchange of step t + 1 (and the owner-side sorts behind it) under step t
                    look = {'t': trn_t.prepare_step(uid, pos_t)}

                    def ahead_step(trn_t=trn_t, pos_t=pos_t, look=look):
                        nxt = trn_t.prepare_step(uid, pos_t)
                        trn_t.training_step(uid, pos_t, ticket=look['t'])
                        look['t'] = nxt
                    with _above_second_stream(True, dev, hi_cache):
                        res_t['ahead'] = timed_max(ahead_step, max(10, args.steps // 4), 3)
                        trn_t.training_step(uid, pos_t, ticket=look['t'])
                    tbl_t.check_overflow()
                on_owners = tbl_t.owner_loss_ok()
                del tbl_t, trn_t
            item_local.copy_(w_keep)
            del w_keep, tower_t
            alg_t = (bytes_per_triplet(d, n, popular) + 2 * 4 * d + 2 * 4 * d / n) * B * n
            extra['train_step'] = {'ms_per_step': round(res_t['sharded'], 4), 'M_triplets_s': round(world * B * n / res_t['sharded'] / 1e3, 2),
                                   'frac_of_hbm_peak_per_gpu': round(alg_t / res_t['sharded'] / 1e6 / HBM_PEAK_GBS, 4),
                                   'world1_reference_ms': round(res_t['world1'], 4),
                                   'efficiency_vs_world1': round(res_t['world1'] / res_t['sharded'], 4),
                                   'loss_on_owners': bool(on_owners),
                                   'one_batch_ahead_ms': round(res_t['ahead'], 4) if 'ahead' in res_t else None,
                                   'what': 'in-place SGD training step (BPR): sample + route, key all-to-all, positives scored by their '
                                           'owners (4-byte-per-query all-reduce), ONE pass over the received rows (scores, loss, query-'
                                           'gradient partials, solo rows updated in place), second small all-reduce, sorted apply for the '
                                           'shared rows, reduce-scatter of the query gradients, user rows'}
        except Exception as e:
            extra['train_step'] = {'error': repr(e)[:300]}
        # the same step with SampledSoftmaxLoss evaluated on the owners (ssm_step_on_owners: two phases around an
        # 8-byte-per-query all-reduce of the queries' (max, sum) partials) and its in-job world-1 reference
        try:
            w_keep = item_local.clone()
            tower_s = torch.nn.Embedding(args.users, d).to(dev)
            with torch.no_grad():
                tower_s.weight.copy_(user)
            res_s = {}
            for tag, pl, rk, smp_s, pos_s, force in (('world1', shard.RowShardPlan(n_loc, 1), 0,
                                                       (ra.PopularSamplerModel(plan.take(counts, rank), lookup=args.pop_lookup).to(dev)
                                                        if popular else ra.UniformSampler(n_loc)), solo_pos, False),
                                                      ('sharded', plan, rank, sampler, pos, True)):
                tbl_s = shard.ShardedItemTable(item_local, pl, rk, dist, check_every=0, force_collectives=force)
                trn_s = shard.ShardedRetriever(tbl_s, tower_s, smp_s, ra.SampledSoftmaxLoss(), n, item_sgd_lr=1e-3, query_sgd_lr=1e-3)
                res_s[tag] = timed_max(lambda trn_s=trn_s, pos_s=pos_s: trn_s.training_step(uid, pos_s), max(10, args.steps // 4), 3)
                tbl_s.check_overflow()
                ssm_own = tbl_s.ssm_owner_ok()
                del tbl_s, trn_s
            item_local.copy_(w_keep)
            del w_keep, tower_s
            extra['train_step_ssm'] = {'ms_per_step': round(res_s['sharded'], 4), 'M_triplets_s': round(world * B * n / res_s['sharded'] / 1e3, 2),
                                       'world1_reference_ms': round(res_s['world1'], 4),
                                       'efficiency_vs_world1': round(res_s['world1'] / res_s['sharded'], 4),
                                       'loss_on_owners': bool(ssm_own),
                                       'what': 'in-place SGD training step (SampledSoftmaxLoss) evaluated on the owners: walk (z, per-query '
                                               'max / sum / weighted-row partials), all-reduce of 8 bytes per query, finish (d, query '
                                               'gradient from the partials, sorted apply of every touched row)'}
        except Exception as e:
            extra['train_step_ssm'] = {'error': repr(e)[:300]}
        # the exact (variable-split, host read-back) exchange
        try:
            exact = shard.ShardedItemTable(item_local, plan, rank, dist, exchange='exact', check_every=0, force_collectives=True)

            def exact_step():
                o = exact.sample_and_score(user, uid, pos, n, sampler)
                return ra.ops.pairwise_loss(nat.LOSS_BPR, o['pos_score'], o['neg_score'], want_grad=True)
            extra['exact_exchange_ms_per_step'] = round(timed_max(exact_step, max(10, args.steps // 4), 5), 4)
        except Exception as e:
            extra['exact_exchange_ms_per_step'] = repr(e)[:200]
        # the single-GPU workload's shape (n = 64, B = 65536: same triplets per GPU per step, 16x larger query gather)
        try:
            n4, b4 = 64, 65536
            g4 = torch.Generator(device=dev).manual_seed(200 + rank)
            uid4 = torch.randint(1, args.users, (b4,), device=dev, generator=g4)
            pos4 = torch.randint(1, args.items, (b4,), device=dev, generator=g4)
            ms4 = timed_max(make_step(table, sampler, uid4, pos4, n4), max(10, args.steps // 4), 5)
            table.check_overflow()
            extra['sharded_n64'] = {'workload': f'same sharded table, neg={n4}, B={b4} queries/step/GPU',
                                    'ms_per_step': round(ms4, 4), 'M_triplets_s': round(world * b4 * n4 / ms4 / 1e3, 2)}
        except Exception as e:
            extra['sharded_n64'] = {'error': repr(e)[:200]}
        # the other sampler on the same sharded table (last: building a 1e8-item popularity table on 1/N of the host
        # cores per rank is the slowest side figure; the watchdog may cut it short without losing the others)
        try:
            other = ra.UniformSampler(args.items).to(dev) if popular else \
                ra.PopularSamplerModel(zipf_counts(args.items, 100_000_000), lookup=args.pop_lookup).to(dev)
            ms_o = timed_max(make_step(table, other, uid, pos, n), max(10, args.steps // 4), 5)
            table.check_overflow()
            extra['other_sampler'] = {'sampler': 'uniform' if popular else 'popular', 'ms_per_step': round(ms_o, 4),
                                      'M_triplets_s': round(world * B * n / ms_o / 1e3, 2)}
            del other
        except Exception as e:
            extra['other_sampler'] = {'error': repr(e)[:200]}
        watchdog.cancel()
        emit()

    if rank == 0 and not emitted:
        print(json.dumps(headline_line(value, ms_step, workload, parallelism, roofline, extra)), flush=True)
    if dist is not None:
        getattr(dist, 'd', dist).destroy_process_group()


if __name__ == '__main__':
    main()
