"""Test / bench harness: a torch.distributed look-alike whose collectives run over gloo through host staging.  Lets
several ranks share ONE GPU (RCCL refuses two ranks on one device), so that the HIP side of the sharded step and the
multi-rank branch of bench.py can be exercised at world size > 1 on a single-GPU box (RSA_BENCH_STAGED=1).  Not part of the
product: the product path is torch.distributed over RCCL."""
import torch


class StagedDist:
    """torch.distributed look-alike whose collectives run over gloo through host staging: lets TWO ranks share ONE
    GPU (RCCL refuses two ranks on one device), so the HIP side of the sharded step -- routing with non-trivial
    split sizes, owner-side scoring of received keys, the sorted backward scatters -- runs at world size 2 on the
    single-GPU test box.  The real RCCL path is the same ShardedItemTable code with torch.distributed itself
    (test_two_gpus_rccl below, skipped without a second GPU)."""

    def __init__(self, dist):
        self.d = dist
        self.ReduceOp = dist.ReduceOp

    def new_group(self, *a, **k):
        return None

    class _Done:
        def wait(self):
            return True

    def all_gather_into_tensor(self, out, x, group=None, async_op=False):
        parts = [torch.empty(x.shape, dtype=x.dtype) for _ in range(self.d.get_world_size())]
        self.d.all_gather(parts, x.cpu())
        out.copy_(torch.cat(parts).view(out.shape))
        return self._Done() if async_op else None

    def all_to_all_single(self, out, x, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
        o = torch.empty(out.shape, dtype=out.dtype)
        self.d.all_to_all_single(o, x.cpu().contiguous(), output_split_sizes, input_split_sizes)
        out.copy_(o)
        return self._Done() if async_op else None

    def reduce_scatter_tensor(self, out, x, group=None):
        full = x.cpu().clone()
        self.d.all_reduce(full)
        r, per = self.d.get_rank(), out.shape[0]
        out.copy_(full[r * per:(r + 1) * per])

    def all_reduce(self, x, op=None, group=None):
        c = x.cpu()
        self.d.all_reduce(c, op=op if op is not None else self.d.ReduceOp.SUM)
        x.copy_(c)

    def barrier(self, group=None):
        self.d.barrier()

    def get_world_size(self, group=None):
        return self.d.get_world_size()

    def get_rank(self, group=None):
        return self.d.get_rank()
