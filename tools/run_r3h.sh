#!/bin/bash
# GPU box, end of round 3: whole GPU suite, every profile again on one box (final kernels), bench line, staged world-2 line
REPO=$(pwd)
OUT=$REPO/gpurun_out/r3h
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_shard.py::test_two_gpus_rccl > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" $OUT/pytest.log | cut -c1-300
timeout 1200 python bench.py > $OUT/bench.log 2> $OUT/bench.err
tail -1 $OUT/bench.log | cut -c1-400
RSA_BENCH_STAGED=1 timeout 600 python bench.py --gpus 2 --items 4000001 --users 100001 --steps 10 --warmup 2 > $OUT/bench_staged.log 2> $OUT/bench_staged.err
tail -1 $OUT/bench_staged.log | cut -c1-300
timeout 2400 bash tools/collect_profiles_r3.sh > $OUT/collect.log 2>&1
grep "^{" gpurun_out/prof_r03/summary.log | cut -c1-160
# the bench command itself under rocprofv3 (stats + FETCH / WRITE / MFMA passes) -> r03_rocprof_summary.txt, r03_pmc_traffic.json
timeout 1500 bash tools/collect_profiles.sh r03 > $OUT/collect_bench.log 2>&1
tail -12 $OUT/collect_bench.log | cut -c1-200
