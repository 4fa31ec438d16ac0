#!/bin/bash
# after the occupancy fixes: GPU suite, bench line, profiles of the shapes whose kernels changed
REPO=$(pwd)
OUT=$REPO/gpurun_out/r3q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_shard.py::test_two_gpus_rccl > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" $OUT/pytest.log | cut -c1-300
timeout 1200 python bench.py > $OUT/bench.log 2> $OUT/bench.err
tail -1 $OUT/bench.log | cut -c1-300
SHAPES="sharded_world1_step sharded_world1_train ssm_N1e6_popular_n256_B8192 sgd_step_N1e7_popular_n64_B65536 train_step_N1e7_popular_n64_B65536" timeout 1500 bash tools/collect_profiles_r3.sh > $OUT/collect.log 2>&1
grep "^{" gpurun_out/prof_r03/summary.log | cut -c1-200
