"""Cost and shape of the forward's output writes on top of the gather + look-up microbenchmark (tools/exp_wr.hip).
Output arrays are taken from SEVERAL allocations: the class of the region they land in is what varies (docs/HISTORY.md 6)."""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, 'libexp_wr.so'))
dev = torch.device('cuda', 0)
N, d = 10_000_001, 128
table = torch.empty(N + 1, d, device=dev).normal_()
B, n = 65536, 64
numel = B * n
ids = torch.randint(1, N, (numel,), device=dev, dtype=torch.int32)
lines = 128 * (1 << 20) // 128
aux = torch.empty(lines * 32, device=dev).normal_()
aux_idx = torch.randint(0, lines, (numel,), device=dev, dtype=torch.int32)
out = torch.zeros(4, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2] * 1e3


def outputs():
    arena = torch.empty(128 << 20, dtype=torch.uint8, device=dev)      # one allocation per output set, like ops.carve
    off = [0]

    def take(nbytes, dtype):
        t = arena[off[0]:off[0] + nbytes].view(dtype)
        off[0] += (nbytes + 4095) // 4096 * 4096
        return t
    o = [take(numel * 8, torch.int64)] + [take(numel * 4, torch.float32) for _ in range(3)] + [take(B * 4, torch.float32) for _ in range(4)]
    return arena, o


names = {0: 'no writes', 1: 'per-element 20 B', 2: 'per-element + 4 per-query floats', 3: 'per-query floats only',
         4: 'per-element staged in LDS (4-tile bursts) + per-query', 5: 'per-element, cached stores', 6: 'per-element, written when the wave has read its last row', 7: 'contiguous chunk of tiles per wave', 8: 'XCD-contiguous tiles'}
warm = torch.empty(1 << 28, device=dev)
t_end = torch.cuda.Event(enable_timing=True)
for _ in range(300):          # ~1.5 s of load: the chip's ramp (docs/HISTORY.md 6, cause 1)
    warm.add_(1.0)
torch.cuda.synchronize()
sets = [outputs() for _ in range(int(os.environ.get("EXP_SETS", "6")))]
for si, (arena, o) in enumerate(sets):
    row = []
    for mode in [int(x) for x in os.environ.get("EXP_MODES", "0,1,2,3,4,5,6,7,8").split(",")]:
        us = timeit(lambda: lib.exp_wr(P(table), P(ids), P(aux), P(aux_idx), ctypes.c_int64(numel), mode, 2048, *[P(t) for t in o], P(out), st))
        row.append(f'{us:7.1f}')
    print(f'output set {si} @ {arena.data_ptr():#x}: ' + ' '.join(row), flush=True)
print('modes: ' + '; '.join(f'{k} = {v}' for k, v in names.items()))
