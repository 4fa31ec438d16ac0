"""Host-side bookkeeping for the device random stream.

The reference draws negatives with ``torch.randint`` / ``torch.rand`` on the
query's device (recstudio/ann/sampler.py:102-104, :246) under
``seed_everything`` (recstudio/utils/utils.py:374-377).  To stay inside that
regime the kernels consume the SAME Philox stream: this module reads
(seed, offset) from the torch device generator, advances the offset by exactly
what the torch call would have, and hands (seed, offset, grid_threads) to the
C ABI.  Sizing rules: PyTorch ``calc_execution_policy``
(ATen/native/hip/DistributionTemplates.h).
"""
from dataclasses import dataclass

import torch

BLOCK = 256


@dataclass(frozen=True)
class PhiloxCall:
    seed: int
    offset: int
    grid_threads: int
    elem_base: int = 0          # element index of this rank's element 0 inside the global call (sharded_stream)


def grid_threads(numel, cu_count, max_threads_per_cu):
    blocks = (int(numel) + BLOCK - 1) // BLOCK
    blocks = min(int(cu_count) * (int(max_threads_per_cu) // BLOCK), blocks)
    return max(blocks, 1) * BLOCK


def counter_offset(numel, g, unroll):
    return ((int(numel) - 1) // (g * unroll) + 1) * 4


def randint_unroll(low, high):
    """torch.randint uses 64-bit draws (2 per philox call) once the range reaches 2**28."""
    return 2 if (int(high) - int(low)) >= (1 << 28) else 4


class _no_shard:
    def __enter__(self):
        self.saved = list(_SHARD)
        _SHARD.clear()

    def __exit__(self, *exc):
        _SHARD.extend(self.saved)


_PROPS = {}


def device_props(device):
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    if idx not in _PROPS:
        p = torch.cuda.get_device_properties(idx)
        _PROPS[idx] = (p.multi_processor_count, p.max_threads_per_multi_processor)
    return _PROPS[idx]


_SHARD = []          # stack of (rank, world, generator) set by sharded_stream


class sharded_stream:
    """``with rng.sharded_stream(rank, world, generator):`` -- inside, a sampler call over ``numel`` outputs draws
    elements [rank*numel, (rank+1)*numel) of ONE torch call over world*numel outputs (grid sized for the global
    call, generator advanced by what the global call consumes).  With ``generator`` in the same state on every rank
    the negatives of a run are those of a single-GPU run on the concatenated batch, whatever the number of GPUs
    ("global element index so results are G-invariant")."""

    def __init__(self, rank, world, generator):
        self.item = (int(rank), int(world), generator)

    def __enter__(self):
        _SHARD.append(self.item)
        return self

    def __exit__(self, *exc):
        _SHARD.pop()


def reserve(numel, unroll, device, generator=None, props=None, repeat=1):
    """Consume the generator state one torch distribution call over ``numel`` outputs would (``repeat`` consecutive such
    calls: the state of the FIRST is returned, call k starts ``k * counter_offset(numel, grid_threads, unroll)`` later)."""
    if numel <= 0:
        return PhiloxCall(0, 0, BLOCK)
    if _SHARD:
        if repeat != 1:
            raise NotImplementedError('reserve(repeat=...) inside sharded_stream')
        rank, world, gen = _SHARD[-1]
        with _no_shard():
            pc = reserve(numel * world, unroll, device, gen if gen is not None else generator, props)
        return PhiloxCall(pc.seed, pc.offset, pc.grid_threads, rank * int(numel))
    if generator is None:
        idx = torch.device(device).index
        if idx is None:
            idx = torch.cuda.current_device()
        if not torch.cuda.is_initialized():
            torch.cuda.init()               # default_generators is empty before the lazy init
        generator = torch.cuda.default_generators[idx]
    cu, mt = props if props is not None else device_props(device)
    g = grid_threads(numel, cu, mt)
    seed = int(generator.initial_seed())
    offset = int(generator.get_offset())
    generator.set_offset(offset + int(repeat) * counter_offset(numel, g, unroll))
    return PhiloxCall(seed & 0xFFFFFFFFFFFFFFFF, offset, g)
