#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/r3e
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -k "sgd_step or graphed or sorted or scatter or Adam or adam" > $OUT/pytest.log 2>&1; grep -E "passed|failed|FAILED|Error" $OUT/pytest.log | head -20
timeout 600 python tools/exp_sgd.py 2>&1 | grep "^{" | tee $OUT/sgd.log
echo "-- 4 waves/SIMD variant of the UPD kernel"
RSA_LIB=$REPO/recstudio_amd/librecstudio_amd_upd4.so timeout 600 python tools/exp_sgd.py 2>&1 | grep "^{" | grep "true" | tee $OUT/sgd_upd4.log
cd /tmp
for v in "1 popular"; do
  name=$(echo $v | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/p_$name -o t -- python $REPO/tools/exp_sgd_prof.py $v > $OUT/p_$name.log 2>&1
done
cd $REPO
python - <<'PY'
import glob, sqlite3
for d in sorted(glob.glob('gpurun_out/r3e/p_1_pop*/')):
    hits = glob.glob(d + '**/*.db', recursive=True)
    if not hits: print(d, 'no db'); continue
    c = sqlite3.connect(hits[0])
    print('==', d)
    for name, calls, total, avg, pct in c.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
        if calls >= 10: print(f'{name[:100]:100s} {calls:5d} {avg:9.1f} us  {pct:5.1f}%')
PY
