import ctypes, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, 'libexp_lut.so')
lib = ctypes.CDLL(so)
dev = torch.device('cuda', 0)
numel = 65536 * 64
out = torch.zeros(4, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
big = torch.empty(10_000_001, 128, device=dev).normal_()        # 5 GB row table to thrash the caches between launches
ids = torch.randint(1, 10_000_001, (numel,), device=dev)
def timeit(fn, reps=20, thrash=False):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for r in range(reps):
        if thrash:
            big[ids[: numel // 4]].sum()        # ~0.5 GB of random row traffic through L2 / MALL
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(r); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]
ids_o = torch.empty(numel, dtype=torch.int64, device=dev)
logp_o = torch.empty(numel, device=dev)
tab2 = torch.rand(1 << 23, 2, device=dev)        # 64 MB of {cdf, prob} pairs
lut23 = torch.rand(1 << 23, 4, device=dev)
for frac in (0, 77, 256):
    for e in (1, 4):
        ms = timeit(lambda r=0: lib.exp_sampler_like(P(lut23), ctypes.c_uint32((1 << 23) - 1), P(tab2), ctypes.c_uint32((1 << 23) - 1), frac, e, ctypes.c_int64(numel), ctypes.c_uint32(r * 7919 + 1), P(ids_o), P(logp_o), st))
        print(f'sampler-like: second-read fraction {frac / 256:.2f} E={e}: {ms * 1e3:8.1f} us', flush=True)
for log2 in (23,):
    entries = 1 << log2
    lut = torch.rand(entries, 4, device=dev)
    for depth in (0,):
        for thrash in (False, True):
            ms = timeit(lambda r=0: lib.exp_lut(P(lut), ctypes.c_uint32(entries - 1), ctypes.c_int64(numel), ctypes.c_uint32(r * 7919 + 1), depth, P(out), st), thrash=thrash)
            print(f'lut 2^{log2} ({entries * 16 / 2**20:7.2f} MB) depth={depth} thrash={int(thrash)}: {ms * 1e3:8.1f} us  {numel * max(depth, 1) / ms / 1e6:8.2f} G reads/s', flush=True)
