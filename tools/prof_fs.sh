#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_fs; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o s -- python $REPO/tools/exp_fs_only.py > $OUT/p$i.log 2>&1
done
cd $REPO
python - <<'PY'
import glob, sqlite3
for db in sorted(glob.glob('gpurun_out/prof_fs/p*/**/*.db', recursive=True)):
    c = sqlite3.connect(db)
    try:
        for name, cn, n, v, dur in c.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%fullscore_kernel%' group by kernel_name, counter_name"):
            print(f'{cn:32s} n={n} avg={v:.4g}  (kernel {dur/1e3:.0f} us)')
    except Exception as e:
        print(db, 'failed', e)
for f in sorted(glob.glob('gpurun_out/prof_fs/p*.log')):
    t = open(f).read()
    if 'rror' in t: print(f, t[-400:])
PY
