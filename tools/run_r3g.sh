#!/bin/bash
# GPU box, round 3: shard tests, re-collection of the profiles whose kernels changed since the first collection, bench line
REPO=$(pwd)
OUT=$REPO/gpurun_out/r3g
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_shard.py -m gpu -q -x --deselect tests/test_gpu_shard.py::test_two_gpus_rccl > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" $OUT/pytest.log | cut -c1-300
SHAPES="${SHAPES:-N1e7_popular_n64_B4096 N1e7_popular_n64_B16384 sgd_step_N1e7_popular_n64_B65536 sharded_world1_step sharded_world1_train}" bash tools/collect_profiles_r3.sh > $OUT/collect.log 2>&1
tail -5 $OUT/collect.log
timeout 1200 python bench.py > $OUT/bench.log 2> $OUT/bench.err
tail -1 $OUT/bench.log
