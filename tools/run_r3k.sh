python -m pytest tests/test_gpu_round2.py tests/test_gpu_retriever.py tests/test_native_abi.py -m gpu -q -x -k "prefetched or fused_optimizer or abi" 2>&1 | tail -4
timeout 900 python bench.py --no-cpu-baseline --no-sweep 2>/dev/null | grep "^{" | python -c "import sys,json; j=json.loads(sys.stdin.read()); t=j['train_step']; print({k:v for k,v in t.items() if k.endswith('_ms')})"
