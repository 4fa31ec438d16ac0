#!/bin/bash
# Run ON THE GPU BOX from the repo root: tools/exp_class_pmc.py under rocprofv3, one --pmc pass per counter group (+ a plain
# kernel trace), then the per-allocation summary -> gpurun_out/class_pmc/class_pmc.json
REPO=$(pwd)
OUT=$REPO/gpurun_out/class_pmc
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
CGROUPS=("" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_THRASHING_STALL_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WR_UNCACHED_32B_sum" "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCC_PROBE_sum")
i=0
for g in "${CGROUPS[@]}"; do
  cd /tmp
  if [ -z "$g" ]; then
    timeout 600 rocprofv3 --kernel-trace -d $OUT/g$i -o p -- python $REPO/tools/exp_class_pmc.py > $OUT/g$i.log 2>&1
  else
    timeout 600 rocprofv3 --kernel-trace --pmc $g -d $OUT/g$i -o p -- python $REPO/tools/exp_class_pmc.py > $OUT/g$i.log 2>&1
  fi
  i=$((i+1))
  cd $REPO
done
python tools/exp_class_pmc.py summarize $OUT && rm -rf $OUT/g[0-9]        # the databases are scratch; class_pmc.json and the logs are kept
