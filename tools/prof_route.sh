#!/bin/bash
# GPU box: kernel times of the routing kernels (tools/exp_route.py), default library and RSA_LIB variants given as args
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for lib in default "$@"; do
  OUT=$REPO/gpurun_out/prof_route_$lib
  rm -rf $OUT; mkdir -p $OUT
  if [ "$lib" = default ]; then unset RSA_LIB; else export RSA_LIB=$REPO/recstudio_amd/librecstudio_amd_$lib.so; fi
  rocprofv3 --kernel-trace --stats -d $OUT/trace -o r -- python $REPO/tools/exp_route.py > $OUT/log.txt 2>&1
  echo "== $lib"; grep "route_fixed" $OUT/log.txt
  python - <<PY
import glob, sqlite3
hits = glob.glob('$OUT/trace/**/*.db', recursive=True)
c = sqlite3.connect(hits[0])
for name, calls, total, avg, pct in c.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
    if 'shard' in name: print(f'{name[:70]:70s} {calls:6d} {avg:10.2f} us')
PY
done
