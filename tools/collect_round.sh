#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: ONE box, one call -- the bench line the driver would record, the same
# command under rocprofv3 (kernel trace + FETCH / WRITE / MFMA counters), rocm-smi before / between / after, then every tracked
# shape alone (tools/collect_shapes.sh).  Everything lands under gpurun_out/; copy the summaries to profiles/.
#   ROUND=r06 bash tools/collect_round.sh
export ROUND=${ROUND:-r06}
mkdir -p gpurun_out/profiles_$ROUND
smi() { rocm-smi --showclocks --showpower --showmaxpower --showperflevel --showcomputepartition --showmemorypartition --showtemp 2>&1 | grep -v "^=\|^$\|WARNING" ; }
{ echo "# before"; smi; } > gpurun_out/profiles_$ROUND/${ROUND}_box_state.txt
python bench.py --steps 20 --warmup 5 2> gpurun_out/bench_round.err | grep "^{" > gpurun_out/profiles_$ROUND/${ROUND}_bench_line.json
{ echo "# after the bench run"; smi; } >> gpurun_out/profiles_$ROUND/${ROUND}_box_state.txt
bash tools/collect_profiles.sh $ROUND > gpurun_out/collect_profiles.log 2>&1
{ echo "# after the rocprofv3 passes of the bench command"; smi; } >> gpurun_out/profiles_$ROUND/${ROUND}_box_state.txt
if [ -z "$SKIP_SHAPES" ]; then bash tools/collect_shapes.sh > gpurun_out/collect_shapes.log 2>&1; fi
{ echo "# after the per-shape collection"; smi; } >> gpurun_out/profiles_$ROUND/${ROUND}_box_state.txt
python - <<'PY'
import json, os
r = os.environ['ROUND']
d = f'gpurun_out/profiles_{r}'
try:
    b = json.loads(open(f'{d}/{r}_bench_line.json').read().strip().splitlines()[-1])
    print('bench:', b['value'], b['ms_per_step'], json.dumps(b['roofline'].get('box')), 'avg_kernel_ms', b['roofline']['avg_kernel_ms'])
except Exception as e:
    print('bench line unreadable', e)
PY
ls gpurun_out/profiles_$ROUND
