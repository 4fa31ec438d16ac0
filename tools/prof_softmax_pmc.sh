#!/bin/bash
# GPU box: PMC passes over tools/exp_softmax_bwd.py (MFMA busy, LDS bank conflicts) per fullscore kernel variant
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pp_$tag
  rocprofv3 --kernel-trace --pmc $pass -d /tmp/pp_$tag -o s -- python $REPO/tools/exp_softmax_bwd.py > /tmp/pp_$tag.log 2>&1
  python - <<PY
import glob, sqlite3, collections
hits = glob.glob('/tmp/pp_$tag/**/*.db', recursive=True)
c = sqlite3.connect(hits[0])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
# rocpd schema: counters_collection view has (kernel name, counter_name, value)
view = [t for t in tabs if 'counters_collection' in t][0]
cols = [r[1] for r in c.execute(f'pragma table_info({view})')]
kn = 'kernel_name' if 'kernel_name' in cols else [x for x in cols if 'name' in x and 'counter' not in x][0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for name, cname, val in c.execute(f'select {kn}, counter_name, value from {view}'):
    if 'fullscore_kernel' in name:
        key = name.split('(')[0][-60:]
        agg[key][cname] += val
for k, d in agg.items():
    print(k, {a: f'{b:.3e}' for a, b in d.items()})
PY
done
