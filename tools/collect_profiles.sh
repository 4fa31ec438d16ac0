#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root.  Collects, under gpurun_out/prof_<tag>/:
#   1. rocprofv3 --kernel-trace --stats of the default bench workload
#   2. two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the same command (fewer steps)
# and writes the summaries tools/summarize_profiles.py turns into profiles/<tag>_*.{txt,json}.
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-sweep"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $BENCH --steps 100 --warmup 10 > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- $BENCH --steps 10 --warmup 2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- $BENCH --steps 10 --warmup 2 > $OUT/pmc_write.log 2>&1
# MFMA utilisation of the full-score GEMM: busy cycles of the matrix pipes vs elapsed GPU cycles
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_mfma -o bench -- $BENCH --steps 4 --warmup 1 > $OUT/pmc_mfma.log 2>&1
cd $REPO
# the rocpd databases are 15-20 MB each (gpurun merges at most 64 MiB back): summarise on the box, keep the summaries
mkdir -p $REPO/gpurun_out/profiles_$TAG
python tools/summarize_profiles.py $OUT $TAG $REPO/gpurun_out/profiles_$TAG > $OUT/summary.log 2>&1
tail -40 $OUT/summary.log
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma
