#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_lut; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc -o s -- python $REPO/tools/exp_lut.py > $OUT/pmc.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $OUT/pmc2 -o s -- python $REPO/tools/exp_lut.py > $OUT/pmc2.log 2>&1
cd $REPO
python - <<'PY'
import glob, sqlite3
for sub in ('pmc', 'pmc2'):
    hits = glob.glob(f'gpurun_out/prof_lut/{sub}/**/*.db', recursive=True)
    if not hits:
        print(sub, 'no db'); print(open(f'gpurun_out/prof_lut/{sub}.log').read()[-600:]); continue
    c = sqlite3.connect(hits[0])
    q = "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"
    for name, cn, n, v in c.execute(q):
        if 'lut_kernel' in name or 'sampler_like' in name or 'chain' in name:
            print(f'{name[:60]:60s} {cn:24s} n={n} avg={v:.0f}  per read (4194304 reads): {v/4194304:.3f}')
PY
