#!/bin/bash
# GPU box: kernel trace of the sharded step at world size 1 with 1 and 2 pipelined query slices
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
for c in 1 2; do
  rm -rf /tmp/pc$c
  CHUNKS=$c rocprofv3 --kernel-trace --stats -d /tmp/pc$c -o s -- python $REPO/tools/exp_shard2.py > /tmp/pc$c.log 2>&1
  echo "== chunks=$c"; grep event_median /tmp/pc$c.log
  python - <<PY
import glob, sqlite3
c = sqlite3.connect(glob.glob('/tmp/pc$c/**/*.db', recursive=True)[0])
tot = 0
for name, calls, total, avg, pct in c.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
    if calls >= 100:
        per_step = total / 1e3 / 105
        tot += per_step
        print(f'{name[:70]:70s} {calls:5d} avg {avg:8.1f}  per-step {per_step:8.1f} us')
print('sum per step', round(tot, 1))
PY
done
