"""Cost of one dependent random look-up per gathered row vs bytes / sectors touched (tools/exp_gran.hip)."""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, 'libexp_gran.so'))
dev = torch.device('cuda', 0)
N, d = 10_000_001, 128
table = torch.empty(N + 1, d, device=dev).normal_()
numel = 65536 * 64
ids = torch.randint(1, N, (numel,), device=dev, dtype=torch.int32)
out = torch.zeros(4, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2]
names = ['none', '4B', '8B', '32B sector', '64B half', 'both halves', 'all 4 sectors']
for mb in (64, 128, 256, 2048):
    lines = mb * (1 << 20) // 128
    aux = torch.empty(lines * 32, device=dev).normal_()
    aux_idx = torch.randint(0, lines, (numel,), device=dev, dtype=torch.int32)
    for mode in range(7):
        ms = timeit(lambda: lib.exp_gran(P(table), P(ids), P(aux), P(aux_idx), ctypes.c_int64(numel), mode, 4096, P(out), st))
        print(f'aux {mb:5d} MB  gather + {names[mode]:14s} {ms*1e3:8.1f} us', flush=True)
    for mode in range(1, 7):
        ms = timeit(lambda: lib.exp_lookup(P(aux), P(aux_idx), ctypes.c_int64(numel), mode, 4096, P(out), st))
        print(f'aux {mb:5d} MB  look-ups alone {names[mode]:14s} {ms*1e3:8.1f} us  {numel/ms/1e6:8.2f} G/s', flush=True)
    del aux, aux_idx
