mkdir -p gpurun_out/r2b
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_shard.py::test_two_gpus_rccl > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
tail -15 gpurun_out/r2b/pytest.log
timeout 600 python bench.py --steps 50 --warmup 10 > gpurun_out/r2b/bench.log 2>&1; echo "bench rc=$?"
grep '^{' gpurun_out/r2b/bench.log | cut -c1-6000
RSA_BENCH_FORCE_SHARD=1 timeout 300 python bench.py --items 12500001 --neg 1024 --batch 4096 --sampler uniform --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r2b/bench_shard1.log 2>&1; echo "shard rc=$?"
grep '^{' gpurun_out/r2b/bench_shard1.log | cut -c1-3000; tail -3 gpurun_out/r2b/bench_shard1.log | cut -c1-400
