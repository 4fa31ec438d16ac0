#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3d
timeout 600 python tools/exp_route2.py 2>&1 | grep "home" | tee gpurun_out/r3d/route2b.log
timeout 900 python -m pytest tests -m gpu -q -x -k "device_loaders or fused_step_object or sample_route or overflow_step or two_ranks_on_one or world1 or config3" 2>&1 | tail -3
