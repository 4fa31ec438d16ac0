"""Where the buffers a launch WRITES live (no reference counterpart: the reference allocates through torch and never looks).

Found in round 5 (profiles/r05_output_placement.json, r05_allocation_class_pmc.json): on MI355X the same launch
-- same kernel, same arguments, same tables -- runs 10-15 % slower when its output arrays live in certain physical regions of
the HBM.  The class is a property of the ALLOCATION (not of the layout inside it), stable for its life, the same from process
to process on one box (consecutive allocations of a fresh process: fast x 5, slow x 11, fast x 4, slow x 16, fast x 4 blocks
of 128 MB), invisible to a pure write stream (which never leaves the 256 MB Infinity Cache) and to every L2 / fabric counter
we could read (same requests, no credit stalls); it shows only under concurrent read load.  The cause is not known.

What this module does about it: ``pick(nbytes, device)`` allocates candidate buffers, times ``rsa_placement_probe`` -- the
forward's access pattern without its arithmetic: random 512-byte rows read from a 1 GiB source, tiles written over the
candidate -- on each, and returns one of the fast class; slow candidates are kept out of circulation (held, up to HELD_CAP
bytes) so that torch's caching allocator cannot hand them back.  Verdicts are cached by address for as long as the allocator
has not returned memory to the driver: a transient allocation that torch recycles is probed once.  ``ops.carve`` routes every output arena of MIN_BYTES or more through here.

**Opt-in** (round 6): ``RSA_PLACEMENT=1`` or ``with placement.enabled():`` switches it on; by default every arena is a plain
``torch.empty``.  Five rounds of driver records showed no gain where it counted (a plain allocation beat the placed one on the
driver's box), the cause of the effect is unknown, and choosing costs a resident 1 GiB probe source, held-back HBM and probe
launches -- not something a training job should pay for unasked.  bench.py times the headline launch on a placed allocation next
to plain ones (``roofline.placed_kernel_ms`` / ``plain_kernel_ms_min``).  Nothing here changes a result: only addresses."""
import contextlib
import ctypes
import os
import time

import torch

from . import _native as nat

ENABLED = os.environ.get('RSA_PLACEMENT', '0') == '1'
MIN_BYTES = 32 << 20          # smaller arenas are not probed (a 25 us probe cannot tell the classes apart)
SOURCE_BYTES = 1 << 30        # read source of the probe: must dwarf the 256 MB Infinity Cache
SPACER_BYTES = 768 << 20      # consecutive allocations are physical neighbours and share a class: skip ahead between candidates
TOLERANCE = 1.06              # the classes are 10-15 % apart, repeats of one class within 1 %, arena sizes within 3 %
MAX_TRIES = 6
BUCKET_BYTES = 16 << 20       # arenas are rounded up to this: a caller whose sizes wander (ragged batches) keeps hitting the same
                              # recycled blocks -- and their cached verdicts -- instead of probing a new size every step
WARM_S = float(os.environ.get('RSA_PLACEMENT_WARM_S', '0.6'))   # seconds of probe launches before a measurement when the GPU may have idled ...
WARM_GAP_S = 0.25             # ... i.e. when the last probe ended longer ago than this
MAX_PROBES = 256              # per device and process: bounds the cost under allocation churn
HELD_CAP = 8 << 30            # bytes of slow-class memory kept out of circulation per device

_state = {}
_off = [0]
_on = [0]


@contextlib.contextmanager
def enabled():
    """Placement inside the block whatever RSA_PLACEMENT says (bench.py: the placed-vs-plain comparison)."""
    _on[0] += 1
    try:
        yield
    finally:
        _on[0] -= 1


def release(dev=None):
    """Give back everything placement holds (the probe source, the held-back slow blocks) on ``dev`` (None: every device)."""
    for idx in list(_state) if dev is None else [dev.index]:
        _state.pop(idx, None)


@contextlib.contextmanager
def disabled():
    """Plain allocations inside the block (bench.py: the same launch on unselected allocations, for comparison)."""
    _off[0] += 1
    try:
        yield
    finally:
        _off[0] -= 1


def _st(dev):
    st = _state.get(dev.index)
    if st is None:
        st = _state[dev.index] = {'source': None, 'best': {}, 'known': {}, 'held': [], 'held_bytes': 0, 'probes': 0,
                                  'picked': 0, 'rejected': 0, 'worst_over_best': 1.0}
    return st


def _epoch(dev):
    """Changes whenever torch's caching allocator gives memory back to the driver (``empty_cache``, an out-of-memory retry): the
    same VIRTUAL address can then come back with other PHYSICAL pages behind it, and a verdict cached by address is stale
    (bench.py's headline inherited the queue figure's "fast" verdict that way and ran on slow memory)."""
    try:
        return int(torch.cuda.memory_stats(dev).get('num_device_free', 0))
    except Exception:
        return 0


def _tiles(nbytes):
    return min(65536, nbytes // 4 // 256 * 256 // 512)


def probe_us(buf, dev=None):
    """Microseconds per launch of the placement probe over ``buf`` (a uint8 tensor of >= 1 MiB, OVERWRITTEN)."""
    dev = buf.device if dev is None else dev
    with torch.cuda.device(dev):          # events and launches on `dev`'s current stream, whatever the caller's current device is
        return _probe_us(buf, dev)


def _probe_us(buf, dev):
    st = _st(dev)
    if st['source'] is None:
        st['source'] = torch.empty(SOURCE_BYTES, dtype=torch.uint8, device=dev)      # (pick() has checked that there is room)
    lib, src = nat.lib(), st['source']
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def launch(salt):
        nat.check(lib.rsa_placement_probe(ctypes.c_void_p(buf.data_ptr()), buf.numel(), ctypes.c_void_p(src.data_ptr()),
                                          src.numel(), salt, stream), 'rsa_placement_probe')
    # the chip runs the same launch 7-11 % slower for about a second after it idled: a probe taken cold would read
    # as the slow class.  Keep the GPU busy with the probe itself first when the last probe was a while ago.
    now = time.perf_counter()
    if now - st.get('last_busy', 0.0) > WARM_GAP_S and torch.cuda.current_stream(dev).query():     # (work pending: not idle)
        t_end = now + WARM_S
        while time.perf_counter() < t_end:
            for w in range(50):
                launch(w)
            torch.cuda.current_stream(dev).synchronize()
    for w in range(2):
        launch(w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream(dev))
    for k in range(3):
        launch(10 + k)
    e1.record(torch.cuda.current_stream(dev))
    e1.synchronize()
    st['probes'] += 1
    st['last_busy'] = time.perf_counter()
    return e0.elapsed_time(e1) / 3 * 1e3


def _ns_per_tile(us, nbytes):
    """The probe's time per tile: 4.99 .. 5.14 ns over 16 K .. 64 K tiles on the fast class (5.7 on the slow one), so ONE
    reference per device serves every arena size."""
    return us * 1e3 / max(_tiles(nbytes), 1)


def pick(nbytes, dev):
    """A uint8 buffer of at least ``nbytes`` (rounded up to BUCKET_BYTES) on ``dev`` from the fast class of allocations (see
    the module docstring); a plain ``torch.empty`` when placement is off, the buffer is small, the stream is being captured,
    the HBM is nearly full or the probe budget is spent."""
    nbytes = int(nbytes)
    if (not (ENABLED or _on[0]) or _off[0] or dev.type != 'cuda' or nbytes < MIN_BYTES or torch.cuda.is_current_stream_capturing()):
        return torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    nbytes = (nbytes + BUCKET_BYTES - 1) // BUCKET_BYTES * BUCKET_BYTES
    st = _st(dev)
    # choosing needs head-room: the probe's 1 GiB read source, candidates, spacers -- checked on EVERY pick (the job may have grown
    # since the last one).  A job that fills the HBM (a 51 GB table with three optimizer states) gets plain allocations, and
    # what placement holds back is released first, rather than an out-of-memory error from a placement attempt.
    free, _ = torch.cuda.mem_get_info(dev)
    need = (0 if st['source'] is not None else SOURCE_BYTES) + MAX_TRIES * (nbytes + SPACER_BYTES) + (4 << 30)
    if free < need:
        release(dev)
        return torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        return _pick(nbytes, dev, st)


def _pick(nbytes, dev, st):
    epoch = _epoch(dev)
    if epoch != st.get('epoch'):
        st['known'].clear()                  # addresses may have been re-backed since the verdicts were taken
        st['epoch'] = epoch
    cands, spacers = [], []
    calibrated = st.get('ref') is not None   # ref: the best ns per tile any candidate on this device ever reached
    for i in range(MAX_TRIES):
        try:
            t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        except torch.cuda.OutOfMemoryError:
            if not cands:
                raise                        # the caller's own allocation does not fit: its error
            break
        key = (t.data_ptr(), nbytes)
        v = st['known'].get(key)
        if v is None:
            if st['probes'] >= MAX_PROBES:
                cands.append((st.get('ref') or 0.0, t))              # budget spent: take it as it comes
                break
            v = st['known'][key] = _ns_per_tile(probe_us(t, dev), nbytes)
        cands.append((v, t))
        st['ref'] = v if st.get('ref') is None else min(st['ref'], v)
        st['best'][_tiles(nbytes)] = min(st['best'].get(_tiles(nbytes), v), v)
        lo, hi = min(c[0] for c in cands), max(c[0] for c in cands)
        if v <= st['ref'] * TOLERANCE and (calibrated or hi > lo * TOLERANCE):
            break            # of the fast class: as good as the best ever seen here and (first time) we have seen a slower one
        if i + 1 < MAX_TRIES:
            try:
                spacers.append(torch.empty(SPACER_BYTES, dtype=torch.uint8, device=dev))
            except torch.cuda.OutOfMemoryError:
                break
    v, chosen = min(cands, key=lambda c: c[0])
    for u, t in cands:
        if t is chosen:
            continue
        if u > st['ref'] * TOLERANCE and st['held_bytes'] + nbytes <= HELD_CAP:
            st['held'].append(t)                     # slow class: never goes back to the caching allocator
            st['held_bytes'] += nbytes
            st['rejected'] += 1
        st['worst_over_best'] = max(st['worst_over_best'], u / max(st['ref'], 1e-9))
    del spacers, cands
    st['picked'] += 1
    return chosen


def summary(dev):
    """Counters for a bench line: probes run, buffers picked / rejected, bytes held back, the largest slow / fast ratio seen."""
    st = _st(dev)
    return {'enabled': bool(ENABLED or _on[0]), 'probes': st['probes'], 'picked': st['picked'], 'rejected_slow': st['rejected'],
            'held_MB': st['held_bytes'] >> 20, 'slowest_over_fastest_probe': round(st['worst_over_best'], 3),
            'probe_ns_per_tile_best': round(st['ref'], 3) if st.get('ref') else None,
            'probe_ns_per_tile_best_by_tiles': {str(k): round(v, 3) for k, v in st['best'].items()}}
