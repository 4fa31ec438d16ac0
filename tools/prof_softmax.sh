#!/bin/bash
# GPU box: rocprofv3 kernel trace of the full-softmax training step parts (tools/exp_softmax_bwd.py) -> profiles_r02/r02_softmax_kernels.txt
REPO=$(pwd); OUT=$REPO/gpurun_out/profiles_r02; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps3
rocprofv3 --kernel-trace --stats -d /tmp/ps3 -o s -- python $REPO/tools/exp_softmax_bwd.py > /tmp/ps3.log 2>&1
cd $REPO
python - <<'PY'
import glob, sqlite3
out = 'gpurun_out/profiles_r02/'
c = sqlite3.connect(glob.glob('/tmp/ps3/**/*.db', recursive=True)[0])
with open(out + 'r02_softmax_kernels.txt', 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats -- python tools/exp_softmax_bwd.py   (full-softmax training step, B = 2048, N = 1e6, d = 128:\n'
            '# forward logsumexp, softmax recompute + write (+ d/d query on the matrix cores), the library GEMMs, autograd steps)\n')
    for l in open('/tmp/ps3.log').read().splitlines():
        if ' ms' in l: f.write('# ' + l + '\n')
    f.write(f'{"kernel":100s} {"calls":>6s} {"avg_us":>10s} {"pct":>6s}\n')
    for name, calls, total, avg, pct in c.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
        f.write(f'{name[:100]:100s} {calls:6d} {avg:10.1f} {pct:6.2f}\n')
print(open(out + 'r02_softmax_kernels.txt').read()[:3000])
PY
