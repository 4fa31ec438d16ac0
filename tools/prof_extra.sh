#!/bin/bash
# GPU box: rocprofv3 kernel traces behind DESIGN.md section 5 (sharded step at world size 1) and section 6 (B = 4096 anatomy).
REPO=$(pwd); OUT=$REPO/gpurun_out/profiles_r02; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps1 /tmp/ps2
rocprofv3 --kernel-trace --stats -d /tmp/ps1 -o s -- python $REPO/tools/exp_shard2.py > /tmp/ps1.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/ps2 -o s -- python $REPO/tools/exp_small.py > /tmp/ps2.log 2>&1
cd $REPO
python - <<'PY'
import glob, sqlite3, collections
out = 'gpurun_out/profiles_r02/'
c = sqlite3.connect(glob.glob('/tmp/ps1/**/*.db', recursive=True)[0])
with open(out + 'r02_shard_kernels.txt', 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats -- python tools/exp_shard2.py   (sharded step at world size 1 over RCCL: 12.5 M-row block,\n'
            '# n = 1024, B = 4096, uniform sampler, fixed-capacity exchange; 5 + 50 + 50 steps)\n')
    f.write('# ' + [l for l in open('/tmp/ps1.log').read().splitlines() if l.startswith('{')][-1] + '\n')
    f.write(f'{"kernel":100s} {"calls":>6s} {"avg_us":>10s} {"pct":>6s}\n')
    for name, calls, total, avg, pct in c.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
        f.write(f'{name[:100]:100s} {calls:6d} {avg:10.1f} {pct:6.2f}\n')
c = sqlite3.connect(glob.glob('/tmp/ps2/**/*.db', recursive=True)[0])
rows = list(c.execute("select name, grid_x, duration from kernels where name like '%fused_fwd_kernel%' order by id"))
with open(out + 'r02_small_batch.txt', 'w') as f:
    f.write('# rocprofv3 --kernel-trace -- python tools/exp_small.py   (B = 4096, n = 64, N = 1e7, fused BPR without the mean; 120 launches\n'
            '# each with the popularity sampler, the uniform sampler, given ids, in that order).  Event-to-event medians (us):\n')
    f.write('# ' + [l for l in open('/tmp/ps2.log').read().splitlines() if l.startswith('{')][-1] + '\n')
    d = [r[2] / 1e3 for r in rows]
    k = len(d) // 3
    for i, tag in enumerate(('popular', 'uniform', 'given')):
        v = sorted(d[i * k:(i + 1) * k])
        f.write(f'kernel duration {tag:8s}: launches {len(v):4d}  median {v[len(v) // 2]:7.2f} us  min {v[0]:7.2f} us\n')
print(open(out + 'r02_shard_kernels.txt').read()[:1800]); print(open(out + 'r02_small_batch.txt').read())
PY
