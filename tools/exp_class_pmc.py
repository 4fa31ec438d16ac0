"""The allocation classes of DESIGN 6 under the hardware counters.  Part 1 (run under `rocprofv3 --kernel-trace --pmc ...`):
NAR separate 128 MB allocations, the headline launch PER launches on each in turn (all nine outputs inside allocation k).
Part 2 (`python tools/exp_class_pmc.py summarize <dir>`): per allocation, the mean kernel duration and the mean of every
collected counter, from the rocpd database -- which counter separates the 412 us allocations from the 470 us ones?"""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == 'summarize':
    out = {}
    for g in sorted(x for x in glob.glob(os.path.join(sys.argv[2], 'g*')) if os.path.isdir(x)):
        meta = None
        for line in open(g + '.log'):
            if line.startswith('{"nar"'):
                meta = json.loads(line)
        dbs = glob.glob(os.path.join(g, '**', '*.db'), recursive=True)
        if not meta or not dbs:
            continue
        c = sqlite3.connect(dbs[0])
        durs = [en - st for name, st, en in c.execute('select name, start, end from kernels order by start') if 'fused_fwd_kernel' in name]
        per = {}
        try:
            for name, counter, value in c.execute('select kernel_name, counter_name, value from counters_collection order by id'):
                if 'fused_fwd_kernel' in name:
                    per.setdefault(counter, []).append(value)
        except sqlite3.Error as e:
            per = {'error': str(e)}
        W, P, NAR = meta['warm'], meta['per'], meta['nar']
        rows = []
        for k in range(NAR):
            lo, hi = W + k * P + 2, W + (k + 1) * P            # skip the first two launches on each allocation
            r = {'arena': k, 'us': round(sum(durs[lo:hi]) / max(1, hi - lo) / 1e3, 1) if len(durs) >= hi else None}
            for cn, vals in per.items():
                if isinstance(vals, list) and len(vals) >= hi:
                    r[cn] = round(sum(vals[lo:hi]) / (hi - lo), 1)
            rows.append(r)
        out[os.path.basename(g)] = {'launches_seen': len(durs), 'expected': W + NAR * P, 'rows': rows}
        rs = sorted((r for r in rows if r['us']), key=lambda r: r['us'])
        print(os.path.basename(g), 'launches', len(durs), 'expected', W + NAR * P)
        for r in rs[:3] + rs[-3:]:
            print('   ', r)
    json.dump(out, open(os.path.join(sys.argv[2], 'class_pmc.json'), 'w'), indent=1)
    sys.exit(0)

import torch                                    # noqa: E402
import recstudio_amd as ra                      # noqa: E402
from recstudio_amd import _native as nat        # noqa: E402
from bench import zipf_counts                   # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
N, U, d, B, n = 10_000_001, 1_000_001, 128, 65536, 64
NAR, PER = int(os.environ.get('NAR', '32')), 8


def table(rows, seed):
    t = torch.empty(rows, d, device=dev).normal_(0, 0.02, generator=torch.Generator(device=dev).manual_seed(seed))
    t[0] = 0
    return t


item, user = table(N, 1), table(U, 2)
ps = ra.PopularSamplerModel(zipf_counts(N, 100_000_000)).to(dev)
gen = torch.Generator(device=dev).manual_seed(100)
uid = torch.randint(1, U, (B,), device=dev, generator=gen)
pos = torch.randint(1, N, (B,), device=dev, generator=gen)
kw = dict(sampler=nat.SAMPLER_POPULAR, query_index=uid, pos_ids=pos, **ps.lookup_kwargs())
arenas = [torch.empty(128 << 20, dtype=torch.uint8, device=dev) for _ in range(NAR)]
SPECS = [('neg_ids', (B, n), torch.int64), ('neg_score', (B, n), torch.float32), ('neg_logp', (B, n), torch.float32),
         ('dneg', (B, n), torch.float32), ('pos_score', (B,), torch.float32), ('pos_logp', (B,), torch.float32),
         ('row_loss', (B,), torch.float32), ('dpos', (B,), torch.float32), ('loss', (), torch.float32)]


def build(a):
    o, off = {}, 0
    for name, shape, dt in SPECS:
        cnt = 1
        for v in shape:
            cnt *= v
        nb = cnt * (8 if dt == torch.int64 else 4)
        o[name] = a[off:off + nb].view(dt).view(shape)
        off += (nb + 4095) // 4096 * 4096
    return o


outs = [build(a) for a in arenas]
warm = 0
for _ in range(int(os.environ.get('WARM', '300'))):        # under --pmc every launch is serialised: count, not time
    ra.ops.fused_forward(item, user, n, out=outs[0], fused_bpr=True, want_mean=False, **kw)
    warm += 1
torch.cuda.synchronize()
for o in outs:
    for _ in range(PER):
        ra.ops.fused_forward(item, user, n, out=o, fused_bpr=True, want_mean=False, **kw)
torch.cuda.synchronize()
print(json.dumps({'nar': NAR, 'per': PER, 'warm': warm, 'va': [hex(a.data_ptr()) for a in arenas]}))
