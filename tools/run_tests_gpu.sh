mkdir -p gpurun_out/tests
python -m pytest tests -m gpu -q --deselect tests/test_gpu_shard.py::test_two_gpus_rccl -x --maxfail=${MAXFAIL:-12} > gpurun_out/tests/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/tests/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/tests/pytest.log | cut -c1-300
