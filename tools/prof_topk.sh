#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_topk; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/t -o s -- python $REPO/tools/exp_topk_trace.py > $OUT/t.log 2>&1
cd $REPO
python - <<'PY'
import glob, sqlite3
c = sqlite3.connect(glob.glob('gpurun_out/prof_topk/t/**/*.db', recursive=True)[0])
for name, calls, total, avg, pct in c.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
    print(f'{name[:100]:100s} {calls:5d} {avg:10.1f} us {pct:6.2f}%')
PY
