"""Kernel timeline of the sharded training step one batch ahead, from a rocprofv3 --kernel-trace database: per HSA queue the
busy time per step, the union of all kernel intervals per step (= time at least one kernel runs) and the time two queues are
busy at once.  usage (on the GPU box):
  rocprofv3 --kernel-trace -d /tmp/ov -o p -- env MODE=ahead STEPS=40 python tools/trace_overlap.py run
  python tools/trace_overlap.py summarize /tmp/ov out.txt"""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch
    import torch.distributed as dist
    import recstudio_amd as ra
    from recstudio_amd import shard
    from recstudio_amd.retriever import _above_second_stream
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29583')
    dist.init_process_group('gloo', rank=0, world_size=1)
    n_blk, n_neg, B, d, U = 12_500_001, 1024, 4096, 128, 1_000_001
    item = torch.empty(n_blk, d, device=dev).normal_(0, 0.02)
    item[0] = 0
    tower = torch.nn.Embedding(U, d).to(dev)
    uid = torch.randint(1, U, (B,), device=dev)
    pos = torch.randint(1, n_blk, (B,), device=dev)
    mode, steps = os.environ.get('MODE', 'ahead'), int(os.environ.get('STEPS', 40))
    tbl = shard.ShardedItemTable(item, shard.RowShardPlan(n_blk, 1), 0, dist, check_every=0)
    tr = shard.ShardedRetriever(tbl, tower, ra.UniformSampler(n_blk), ra.BPRLoss(), n_neg, item_sgd_lr=1e-3, query_sgd_lr=1e-3,
                                overlap_query_rows=mode != 'serial')
    with _above_second_stream(mode == 'ahead', dev, {}):
        if mode == 'ahead':
            tk = tr.prepare_step(uid, pos)
            for i in range(steps):
                nxt = tr.prepare_step(uid, pos)
                tr.training_step(uid, pos, ticket=tk)
                tk = nxt
            tr.training_step(uid, pos, ticket=tk)
        else:
            for i in range(steps):
                tr.training_step(uid, pos)
    torch.cuda.synchronize()
    print(json.dumps({'mode': mode, 'steps': steps}))


def summarize(d, out):
    lines = []
    for mode in sorted(os.listdir(d)):
        dbs = glob.glob(os.path.join(d, mode, '**', '*.db'), recursive=True)
        if not dbs:
            continue
        c = sqlite3.connect(dbs[0])
        rows = [(q, s, e, n) for q, s, e, n in c.execute('select queue_id, start, end, name from kernels order by start') if 'rsa::' in n]
        walks = [r for r in rows if 'owner_backward_walk' in r[3]]
        if len(walks) < 12:
            continue
        # steps 5 .. last-2: from the start of a walk to the start of the next
        t0, t1, nsteps = walks[5][1], walks[-2][1], len(walks) - 7
        sel = [r for r in rows if t0 <= r[1] < t1]
        by_q = {}
        for q, s, e, _ in sel:
            by_q.setdefault(q, []).append((s, e))
        ev = sorted([(s, 1) for _, s, e, _ in sel] + [(e, -1) for _, s, e, _ in sel])
        depth, last, any_busy, two_busy = 0, None, 0, 0
        for t, dlt in ev:
            if last is not None:
                if depth >= 1:
                    any_busy += t - last
                if depth >= 2:
                    two_busy += t - last
            depth += dlt
            last = t
        lines.append(f'## {mode}: {nsteps} steps, wall {(t1 - t0) / nsteps / 1e3:.1f} us per step (walk start to walk start)')
        for q, iv in sorted(by_q.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
            lines.append(f'queue {q}: {len(iv) / nsteps:5.1f} kernels per step, busy {sum(e - s for s, e in iv) / nsteps / 1e3:8.1f} us per step')
        lines.append(f'sum of kernel durations {sum(e - s for _, s, e, _ in sel) / nsteps / 1e3:.1f} us per step; some kernel running '
                     f'{any_busy / nsteps / 1e3:.1f} us; two or more at once {two_busy / nsteps / 1e3:.1f} us; idle '
                     f'{((t1 - t0) - any_busy) / nsteps / 1e3:.1f} us')
        lines.append('')
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run()
    else:
        summarize(sys.argv[2], sys.argv[3])
