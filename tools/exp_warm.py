"""The headline launch timed against HOW LONG the GPU has been kept busy (same process, same allocation): the kernel is ~10 %
faster once the chip has been under load for several seconds -- the spread round 4 put down to "boxes".  Prints the ramp and,
cold / warm, every clock amd-smi / rocm-smi report under load.  `python tools/exp_warm.py`"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recstudio_amd as ra                      # noqa: E402
from recstudio_amd import _native as nat        # noqa: E402
from bench import zipf_counts                   # noqa: E402

dev = torch.device('cuda', 0)
N, U, d, B, n = 10_000_001, 1_000_001, 128, 65536, 64
gen = torch.Generator(device=dev).manual_seed(100)
uid = torch.randint(1, U, (B,), device=dev, generator=gen)
pos = torch.randint(1, N, (B,), device=dev, generator=gen)
ps = ra.PopularSamplerModel(zipf_counts(N, 100_000_000)).to(dev)
g = torch.Generator(device=dev).manual_seed(1)
item = torch.empty(N, d, device=dev).normal_(0, 0.02, generator=g)
user = torch.empty(U, d, device=dev).normal_(0, 0.02, generator=g)
kw = dict(query_index=uid, pos_ids=pos, sampler=nat.SAMPLER_POPULAR, **ps.lookup_kwargs())
buf = {}


def launch():
    buf['o'] = ra.ops.fused_forward(item, user, n, out=buf.get('o'), fused_bpr=True, want_mean=False, **kw)


def timed(k=40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / k * 1e3, 1)


def clocks_under_load():
    stop = threading.Event()

    def spin():
        torch.cuda.set_device(0)
        while not stop.is_set():
            for _ in range(50):
                launch()
            torch.cuda.synchronize()
    th = threading.Thread(target=spin, daemon=True)
    th.start()
    time.sleep(0.3)
    out = {}
    for cmd in (['amd-smi', 'metric', '-c', '-p', '--json'], ['rocm-smi', '--showclocks', '--showpower', '--showtemp', '--json']):
        try:
            out[cmd[0]] = subprocess.run(cmd, capture_output=True, text=True, timeout=30).stdout[-3000:]
        except Exception as e:
            out[cmd[0]] = repr(e)
    stop.set()
    th.join()
    return out


for _ in range(5):
    launch()
torch.cuda.synchronize()
time.sleep(3.0)                                  # idle first: start from the state a fresh process sees
res = {'ramp_us_at_busy_seconds': []}
t0 = time.perf_counter()
res['ramp_us_at_busy_seconds'].append([0.0, timed(20)])
cold = clocks_under_load()
marks = [0.5, 1, 2, 3, 4, 6, 8, 12, 16, 20]
for m in marks:
    while time.perf_counter() - t0 < m:
        for _ in range(50):
            launch()
        torch.cuda.synchronize()
    res['ramp_us_at_busy_seconds'].append([round(time.perf_counter() - t0, 2), timed(20)])
warm = clocks_under_load()
time.sleep(5.0)
res['after_5s_idle_us'] = timed(20)
time.sleep(0.02)
res['then_us'] = [timed(20) for _ in range(3)]
res['cold_clocks'] = cold
res['warm_clocks'] = warm
print(json.dumps(res))
