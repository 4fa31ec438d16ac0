#!/bin/bash
# In-box A/B of library builds (tools/build_variant.sh) on one tracked shape: every variant runs tools/prof_shapes.py alone under
# rocprofv3 --kernel-trace --stats; prints the event time per step and the step's top kernels per variant.
#   SHAPE=sharded_world1_train_ssm VARIANTS="default allnt old" bash tools/exp_variants.sh
SHAPE=${SHAPE:-sharded_world1_train_ssm}
VARIANTS=${VARIANTS:-default}
K=${K:-60}
REPO=$(pwd)
OUT=$REPO/gpurun_out/variants_$SHAPE
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for v in $VARIANTS; do
  lib=$REPO/recstudio_amd/librecstudio_amd_$v.so
  [ "$v" = default ] && lib=$REPO/recstudio_amd/librecstudio_amd.so
  rm -rf $OUT/$v
  RSA_LIB=$lib PROF_WARM_MS=${PROF_WARM_MS:-1500} timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/$v -o p -f csv -- python $REPO/tools/prof_shapes.py $SHAPE $K > $OUT/$v.log 2>&1
  echo "== $v: $(grep -o '"event_us_per_step": [0-9.]*' $OUT/$v.log)"
  python - $OUT/$v $K <<'PY'
import csv, glob, sys
d, K = sys.argv[1], int(sys.argv[2])
f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:8]:
    print(f"   {r['Name'][:90]:90s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f}")
PY
  rm -rf $OUT/$v
done
