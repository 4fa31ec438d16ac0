#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_train; rm -rf $OUT; mkdir -p $OUT $REPO/gpurun_out/profiles_r02
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/t -o s -- python $REPO/tools/exp_train_trace.py > $OUT/t.log 2>&1
cd $REPO
python - <<'PY'
import glob, sqlite3
c = sqlite3.connect(glob.glob('gpurun_out/prof_train/t/**/*.db', recursive=True)[0])
with open('gpurun_out/profiles_r02/r02_train_kernels.txt', 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats -- python tools/exp_train_trace.py   (B=65536, n=64, d=128, N=1e7, popularity sampler:\n'
            '# 12 in-place SGD steps + 12 lazy-Adam steps + 6 autograd steps with dense gradients)\n')
    f.write(f'{"kernel":100s} {"calls":>6s} {"avg_us":>10s} {"pct":>6s}\n')
    for name, calls, total, avg, pct in c.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
        line = f'{name[:100]:100s} {calls:6d} {avg:10.1f} {pct:6.2f}'
        print(line); f.write(line + '\n')
PY
rm -rf $OUT/t
