import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, 'libexp.so'))
dev = torch.device('cuda', 0)
N, d = 10_000_001, 128
table = torch.empty(N, d, device=dev).normal_()
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
    return t[len(t)//2]
gb = numel * 512 / 1e9
names = ['U4', 'U8', 'U16', 'U32', 'U8nt', 'U16nt']
for v in range(6):
    for blocks, threads in ((2048, 256), (4096, 256), (1024, 256), (8192, 64), (16384, 64), (2048, 512), (1024, 1024)):
        ms = timeit(lambda: lib.exp_gather(P(table), P(ids), ctypes.c_int64(numel), v, blocks, threads, P(out), st))
        print(f'{names[v]:6s} blocks={blocks:6d} threads={threads:5d}  {ms*1e3:8.1f} us  {gb/ms*1e3:8.1f} GB/s', flush=True)
n4 = numel * 32
for blocks in (2048, 4096, 8192):
    ms = timeit(lambda: lib.exp_stream(P(table), ctypes.c_int64(n4), blocks, P(out), st))
    print(f'stream blocks={blocks}  {ms*1e3:8.1f} us  {gb/ms*1e3:8.1f} GB/s')
# sorted ids (DRAM page locality)
ids_sorted = ids.sort().values.contiguous()
for v in (1, 2):
    ms = timeit(lambda: lib.exp_gather(P(table), P(ids_sorted), ctypes.c_int64(numel), v, 2048, 256, P(out), st))
    print(f'{names[v]} sorted ids: {ms*1e3:8.1f} us  {gb/ms*1e3:8.1f} GB/s')
