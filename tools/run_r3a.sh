#!/bin/bash
# GPU box, round 3 first call: the v2 sharded exchange -- tests, step timings, per-kernel profile
REPO=$(pwd)
OUT=$REPO/gpurun_out/r3a
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_shard.py tests/test_gpu_round2.py -x -q -k "shard or config3 or route or rccl or two_ranks or launch" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
for cfg in "STEP=fwd" "STEP=old" "STEP=fwd CHUNKS=2" "STEP=fwd CHUNKS=4" "STEP=fwd SAMPLER=popular" "STEP=train" "STEP=fwd NEG=64 B=65536"; do
  echo "== $cfg" >> $OUT/steps.log
  env $cfg timeout 300 python tools/exp_shard2.py >> $OUT/steps.log 2>&1
done
cat $OUT/steps.log
cd /tmp
for cfg in "STEP=fwd" "STEP=train"; do
  name=$(echo $cfg | tr ' =' '__')
  env $cfg timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$name -o t -- python $REPO/tools/exp_shard2.py > $OUT/prof_$name.log 2>&1
done
cd $REPO
python - <<'PY'
import glob, sqlite3, os
for d in sorted(glob.glob('gpurun_out/r3a/prof_*/')):
    hits = glob.glob(d + '**/*.db', recursive=True)
    if not hits:
        print(d, 'no db'); continue
    c = sqlite3.connect(hits[0])
    with open(d.rstrip('/') + '_top.txt', 'w') as f:
        for name, calls, total, avg, pct in c.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
            line = f'{name[:110]:110s} {calls:6d} {total/1e3:10.3f} {avg:10.2f} {pct:6.2f}'
            f.write(line + '\n')
    print(open(d.rstrip('/') + '_top.txt').read()[:3000])
PY
