#!/bin/bash
# Run ON THE GPU BOX from the repo root: the L2 <-> fabric REQUEST counters of the dominant kernel of a few figures
# (tools/prof_shapes.py), one --pmc pass per counter group (PMC passes carry --kernel-trace only) -> one JSON + text summary
# under gpurun_out/profiles_$ROUND/${ROUND}_requests.{json,txt}.  Evidence for the request-rate model of DESIGN 4.1 (the gather is
# bound by 128-byte requests per second, not by bytes).
export ROUND=${ROUND:-r04}
SHAPES=${SHAPES:-"headline_N1e7_popular_n64_B65536 N1e8_uniform_n64_B65536 N1e8_popular_n64_B65536 sharded_world1_train"}
REPO=$(pwd)
OUT=$REPO/gpurun_out/req_$ROUND
DST=$REPO/gpurun_out/profiles_$ROUND
mkdir -p $OUT $DST
export TMPDIR=/tmp
CGROUPS=("TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum")
for s in $SHAPES; do
  i=0
  for g in "${CGROUPS[@]}"; do
    cd /tmp
    timeout 900 rocprofv3 --kernel-trace --pmc $g -d $OUT/$s/g$i -o p -- env PROF_WARM_MS=0 python $REPO/tools/prof_shapes.py $s 8 > $OUT/$s.g$i.log 2>&1
    i=$((i+1))
  done
  cd $REPO
done
python tools/summarize_requests.py $OUT $DST "$SHAPES"
rm -rf $OUT/*/g*          # the databases are scratch (gpurun_out/ is capped at 64 MiB); the summary is what is kept
