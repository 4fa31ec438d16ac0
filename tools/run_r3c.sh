REPO=$(pwd)
OUT=$REPO/gpurun_out/r3c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --maxfail=8 --deselect tests/test_gpu_shard.py::test_two_gpus_rccl -k "fit_two_staged or config2_sasrec or apply_rows or long_queries or sample_route or overflow_step or seq or sasrec or SASRec" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" $OUT/pytest.log | cut -c1-300
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|socket.cpp\|amdgpu.ids" $OUT/pytest.log | tail -40
