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


def grid_threads(numel, cu_count, max_threads_per_cu):
    blocks = (int(numel) + BLOCK - 1) // BLOCK
    blocks = min(int(cu_count) * (int(max_threads_per_cu) // BLOCK), blocks)
    return max(blocks, 1) * BLOCK


def counter_offset(numel, g, unroll):
    return ((int(numel) - 1) // (g * unroll) + 1) * 4


def randint_unroll(low, high):
    """torch.randint uses 64-bit draws (2 per philox call) once the range reaches 2**28."""
    return 2 if (int(high) - int(low)) >= (1 << 28) else 4


_PROPS = {}


def device_props(device):
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    if idx not in _PROPS:
        p = torch.cuda.get_device_properties(idx)
        _PROPS[idx] = (p.multi_processor_count, p.max_threads_per_multi_processor)
    return _PROPS[idx]


def reserve(numel, unroll, device, generator=None, props=None):
    """Consume the generator state one torch distribution call over ``numel`` outputs would."""
    if numel <= 0:
        return PhiloxCall(0, 0, BLOCK)
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
    generator.set_offset(offset + counter_offset(numel, g, unroll))
    return PhiloxCall(seed & 0xFFFFFFFFFFFFFFFF, offset, g)
