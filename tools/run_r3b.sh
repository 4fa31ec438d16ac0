#!/bin/bash
# GPU box, round 3: whole GPU test suite, sharded step timings, the bench line, the staged world-2 bench line
REPO=$(pwd)
OUT=$REPO/gpurun_out/r3b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --maxfail=8 --deselect tests/test_gpu_shard.py::test_two_gpus_rccl > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" $OUT/pytest.log | cut -c1-300
tail -40 $OUT/pytest.log | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -30
: > $OUT/steps.log
for cfg in "STEP=fwd" "STEP=fwd CHUNKS=2" "STEP=fwd SAMPLER=popular" "STEP=train" "STEP=fwd NEG=64 B=65536"; do
  echo "== $cfg" >> $OUT/steps.log
  env $cfg timeout 300 python tools/exp_shard2.py 2>&1 | grep "^{" >> $OUT/steps.log
done
cat $OUT/steps.log
timeout 1200 python bench.py > $OUT/bench.log 2> $OUT/bench.err
tail -1 $OUT/bench.log
RSA_BENCH_STAGED=1 timeout 600 python bench.py --gpus 2 --items 4000001 --users 100001 --steps 10 --warmup 2 > $OUT/bench_staged.log 2> $OUT/bench_staged.err
tail -1 $OUT/bench_staged.log
tail -5 $OUT/bench_staged.err
