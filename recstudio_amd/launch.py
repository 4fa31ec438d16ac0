"""Multi-GPU training entry point: one process per GPU, item table row-sharded, query tower replicated.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m recstudio_amd.launch --items 100000001 --users 1000001 --dim 128 --neg 1024 --batch 4096 --steps 100

This is what replaces recstudio/utils/data_parallel.py (which re-broadcasts every parameter of the model to every
device on every step, :106-159) and the dead DDP branch (recommender.py:731-740): rank r owns rows
[r*rows_per_shard, ...) of the item table (and of its optimizer state), a data-parallel slice of the batch and a
replica of the user tower.  Per step (`shard.ShardedRetriever.training_step`, BPR evaluated on the owners of the
negatives): sample -> route -> RCCL all-to-all of 8-byte keys; the positives scored by their owners (4-byte-per-query
all-reduce); ONE pass over the received rows (scores, loss, query-gradient partials, rows updated in place); a second
small all-reduce; sorted apply of the shared rows; reduce-scatter of the query gradients.  No score crosses the fabric,
the item rows never leave their owner; the replicated user table's row-sparse gradient is all-gathered as
(ids, rows) and applied in place on every replica (an MLP tower's dense gradients would be summed with one bucketed
all-reduce).  Synthetic interactions (there is no dataset download in scope); the loop is plain SGD.
"""
import argparse
import os
import time

import torch


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--items', type=int, default=10_000_001)
    ap.add_argument('--users', type=int, default=1_000_001)
    ap.add_argument('--dim', type=int, default=128)
    ap.add_argument('--neg', type=int, default=64)
    ap.add_argument('--batch', type=int, default=65536, help='queries per step PER GPU')
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--lr', type=float, default=0.05)
    ap.add_argument('--seed', type=int, default=2022)
    ap.add_argument('--layout', choices=('block', 'interleaved'), default='block',
                    help="row ownership: contiguous blocks, or rows r, r + G, ... (no hot shard when ids are ordered by popularity)")
    args = ap.parse_args(argv)

    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29511')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)     # "nccl" is RCCL on ROCm
    try:
        import recstudio_amd as ra
        from recstudio_amd.shard import RowShardPlan, ShardedItemTable, ShardedRetriever
        ra._native.lib()
        plan = RowShardPlan(args.items, world, layout=args.layout)
        g = torch.Generator(device=dev).manual_seed(args.seed + 17 * rank)
        item_local = torch.empty(plan.n_local(rank), args.dim, device=dev).normal_(0, 0.02, generator=g)   # init.py:18-27
        if rank == 0:
            item_local[0] = 0                                                                 # padding row
        # the user tower is replicated: identical initial weights on every rank
        user = torch.nn.Embedding(args.users, args.dim, padding_idx=0).to(dev)
        with torch.no_grad():
            user.weight.normal_(0, 0.02, generator=torch.Generator(device=dev).manual_seed(args.seed))
            user.weight[0] = 0
        torch.manual_seed(args.seed + rank)
        # negatives come from ONE job-wide Philox stream (same seed on every rank, global element index): the run's
        # negatives do not depend on the number of GPUs
        table = ShardedItemTable(item_local, plan, rank, dist, sample_seed=args.seed)
        inplace = args.dim in (64, 128, 256)       # item rows updated inside the backward exchange (no dense gradient block)
        # the user table is an nn.Embedding: its gradient travels as (ids, rows) -- two small all-gathers -- and is
        # applied in place on every replica (ShardedRetriever, sparse_query_rows); no dense [users, d] gradient exists
        trainer = ShardedRetriever(table, user, ra.UniformSampler(args.items), ra.BPRLoss(), args.neg,
                                   item_sgd_lr=args.lr if inplace else None, query_sgd_lr=args.lr)
        data = torch.Generator(device=dev).manual_seed(args.seed + 1000 + rank)
        t0, losses = None, []
        for step in range(args.steps):
            if step == min(5, args.steps - 1):
                torch.cuda.synchronize()
                dist.barrier()
                t0, s0 = time.perf_counter(), step
            uid = torch.randint(1, args.users, (args.batch,), device=dev, generator=data)
            # every user keeps consuming the same handful of items: something for the model to learn
            pos = 1 + (uid * 2654435761 + torch.randint(0, 4, (args.batch,), device=dev, generator=data)) % (args.items - 1)
            if not inplace:
                trainer.item_grad_local.zero_()
            loss = trainer.training_step(uid, pos)        # this rank's share of the global mean loss
            if not inplace:
                with torch.no_grad():                     # plain SGD on the owned item rows
                    item_local.add_(trainer.item_grad_local, alpha=-args.lr)
            if step % 10 == 0 or step == args.steps - 1:
                total = loss.clone()
                dist.all_reduce(total)
                losses.append(float(total))
                if rank == 0:
                    print(f'step {step:5d}  loss {float(total):.5f}', flush=True)
        torch.cuda.synchronize()
        dist.barrier()
        if rank == 0 and t0 is not None and args.steps - s0 > 0:
            dt = (time.perf_counter() - t0) / (args.steps - s0)
            print(f'{world} GPU(s): {dt * 1e3:.3f} ms/step, {world * args.batch * args.neg / dt / 1e6:.1f} M triplets/s '
                  f'(forward + backward + SGD)', flush=True)
        return losses
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
