#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_sampler; rm -rf $OUT; mkdir -p $OUT
python tools/exp_sampler.py 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc -o s -- python $REPO/tools/exp_sampler.py > $OUT/pmc.log 2>&1
cd $REPO
python - <<'PY'
import glob, sqlite3
c = sqlite3.connect(glob.glob('gpurun_out/prof_sampler/pmc/**/*.db', recursive=True)[0])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
q = "select kernel_name, count(*), avg(value) from counters_collection where counter_name='FETCH_SIZE' group by kernel_name"
try:
    for name, n, v in c.execute(q):
        print(f'{name[:80]:80s} n={n} FETCH_SIZE avg={v:.0f} KB-units -> x2 corrected {v*2*1024/1e6:.1f} MB  per id {v*2*1024/4194304:.1f} B')
except Exception as e:
    print('query failed', e); print(tabs)
    print([r for r in c.execute("pragma table_info(counters_collection)")])
PY
