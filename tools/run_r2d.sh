mkdir -p gpurun_out/r2d
python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -4
timeout 200 python tools/exp_r2.py small 2>&1 | grep RESULT | cut -c1-600
timeout 120 python bench.py --gpus 2 --steps 3 --warmup 1 --no-sweep --no-cpu-baseline > gpurun_out/r2d/gpus2.log 2>&1; echo "gpus2 rc=$?"; grep -E "Error|error|rank|invalid" gpurun_out/r2d/gpus2.log | head -8 | cut -c1-300
