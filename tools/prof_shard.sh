#!/bin/bash
# GPU box: per-kernel times of the sharded step (world_size 1 over RCCL) -> gpurun_out/prof_shard/top.txt
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_shard
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RSA_BENCH_FORCE_SHARD=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $REPO/bench.py --no-cpu-baseline --no-sweep --steps 50 --warmup 5 > $OUT/trace.log 2>&1
cd $REPO
python - <<'PY'
import glob, sqlite3
hits = glob.glob('gpurun_out/prof_shard/trace/**/*.db', recursive=True)
c = sqlite3.connect(hits[0])
with open('gpurun_out/prof_shard/top.txt', 'w') as f:
    for name, calls, total, avg, pct in c.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
        line = f'{name[:110]:110s} {calls:6d} {total/1e3:10.3f} {avg:10.2f} {pct:6.2f}'
        print(line); f.write(line + '\n')
PY
tail -3 $OUT/trace.log
