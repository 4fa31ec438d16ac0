#!/bin/bash
# GPU box: kernel trace of the sharded TRAINING step (recstudio_amd.launch at world size 1), args passed through
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/plaunch
rocprofv3 --kernel-trace --stats -d /tmp/plaunch -o s -- python -m recstudio_amd.launch "$@" > /tmp/plaunch.log 2>&1
grep "GPU(s)" /tmp/plaunch.log
python - <<'PY'
import glob, sqlite3
c = sqlite3.connect(glob.glob('/tmp/plaunch/**/*.db', recursive=True)[0])
for name, calls, total, avg, pct in c.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
    if pct > 0.4: print(f'{name[:96]:96s} {calls:5d} avg {avg:9.1f} us {pct:6.2f} %')
rows = [(r[0], r[1] / 1e3) for r in c.execute("select grid_x, duration from kernels where name like '%sorted_apply%' order by id")]
print('sorted_apply calls of the last two steps (grid threads, us):', [(g, round(d, 1)) for g, d in rows[-6:]])
PY
