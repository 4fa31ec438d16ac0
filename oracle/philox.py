"""Oracle (test infrastructure): the device random stream the reference draws from.

The reference samples negatives with ``torch.randint`` / ``torch.rand`` on the
query's device (``recstudio/ann/sampler.py:102-104`` and ``:246``).  On a ROCm
device those calls run PyTorch's ``distribution_elementwise_grid_stride_kernel``
over rocRAND's Philox4x32-10 engine.  Neither PyTorch nor rocRAND is vendored
under /root/reference, so the published algorithm is restated here:

* Philox4x32-10: Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"
  (Random123).  Pinned by the Random123 known-answer vectors
  (tests/test_oracle_golden.py::test_philox_kat).
* Element -> (subsequence, counter, component) mapping: PyTorch 2.10
  ``ATen/native/hip/DistributionTemplates.h`` -- ``calc_execution_policy``
  (grid = min(CUs * (maxThreadsPerCU / 256), ceil(numel / 256)) blocks of 256),
  ``hiprand_init(seed, thread_idx, offset)``, grid-stride loop with unroll 4
  (32-bit integers, floats) or 2 (64-bit integers when range >= 2**28), element
  ``li = idx + G*(unroll*k + ii)`` takes component ``ii`` of the k-th draw.
* int transform: ``val % range + base`` (ATen/core/TransformationHelper.h:42).
* float transform: rocRAND ``uniform_distribution``: ``2^-32 + v * 2^-32`` in
  fp32 (value in (0, 1]), then torch's ``value == 1 ? 0 : value`` flip.

Pinned against the real thing on the GPU box by
tests/test_gpu_parity.py::test_device_stream_matches_torch.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: uint32 array [..., 4]; key: uint32 array [..., 2] (broadcastable).
    Returns uint32 [..., 4]."""
    ctr = np.asarray(ctr, dtype=np.uint64)
    key = np.asarray(key, dtype=np.uint64)
    c0, c1, c2, c3 = (ctr[..., i].copy() for i in range(4))
    k0 = key[..., 0].copy()
    k1 = key[..., 1].copy()
    for _ in range(10):
        p0 = M0 * c0          # < 2^64, exact in uint64
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & MASK32, lo1, (hi0 ^ c3 ^ k1) & MASK32, lo0
        k0 = (k0 + np.uint64(W0)) & MASK32
        k1 = (k1 + np.uint64(W1)) & MASK32
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def rng_grid_threads(numel, cu_count, max_threads_per_cu, block=256):
    """Total threads G of torch's distribution kernel for ``numel`` outputs."""
    blocks = (int(numel) + block - 1) // block
    blocks = min(int(cu_count) * (int(max_threads_per_cu) // block), blocks)
    return blocks * block


def rng_counter_offset(numel, grid_threads, unroll):
    """Amount torch advances the generator's philox offset for one call."""
    return ((int(numel) - 1) // (grid_threads * unroll) + 1) * 4


def _draw(seed, offset, numel, grid_threads, unroll):
    """uint32 [numel, 4] philox output and the component index each element uses."""
    li = np.arange(int(numel), dtype=np.uint64)
    G = np.uint64(grid_threads)
    idx = li % G
    j = li // G
    k = j // np.uint64(unroll)
    ii = (j % np.uint64(unroll)).astype(np.int64)
    c = np.uint64(int(offset) // 4) + k
    ctr = np.stack([c & MASK32, c >> np.uint64(32), idx & MASK32, idx >> np.uint64(32)], axis=-1)
    key = np.array([int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    out = philox4x32_10(ctr, key)
    return out, ii


def device_randint(seed, offset, numel, low, high, grid_threads):
    """== torch.randint(low, high, (numel,), device='cuda') for generator state
    (seed, offset) on a device whose distribution grid has ``grid_threads``."""
    rng = np.uint64(int(high) - int(low))
    if int(rng) >= (1 << 28):
        out, ii = _draw(seed, offset, numel, grid_threads, 2)
        o = out.astype(np.uint64)
        v0 = (o[:, 0] << np.uint64(32)) | o[:, 1]
        v1 = (o[:, 2] << np.uint64(32)) | o[:, 3]
        v = np.where(ii == 0, v0, v1)
    else:
        out, ii = _draw(seed, offset, numel, grid_threads, 4)
        v = out[np.arange(int(numel)), ii].astype(np.uint64)
    return (v % rng).astype(np.int64) + np.int64(low)


def device_rand(seed, offset, numel, grid_threads):
    """== torch.rand(numel, device='cuda') (fp32) for generator state (seed, offset)."""
    out, ii = _draw(seed, offset, numel, grid_threads, 4)
    v = out[np.arange(int(numel)), ii]
    inv = np.float32(2.3283064e-10)          # 2^-32
    u = (v.astype(np.float32) * inv + inv).astype(np.float32)
    u = np.where(u == np.float32(1.0), np.float32(0.0), u)
    return u.astype(np.float32)
