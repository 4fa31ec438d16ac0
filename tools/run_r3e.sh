#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3e
timeout 900 python -m pytest tests -m gpu -q -x -k "sgd_step or graphed" > gpurun_out/r3e/pytest.log 2>&1; grep -E "passed|failed|FAILED|Error|error" gpurun_out/r3e/pytest.log | head -20
timeout 600 python tools/exp_sgd.py 2>&1 | grep "^{" | tee gpurun_out/r3e/sgd.log
