#!/bin/bash
# headline-size A/B: transposed fold (default above 32768 tiles) vs the butterfly tile everywhere, alternating processes
export TMPDIR=/tmp
REPO=$(pwd)
mkdir -p gpurun_out/r3f
python - <<'PY' > /tmp/ps.log 2>&1
import sys, torch
sys.path.insert(0, '.')
import recstudio_amd as ra
from bench import zipf_counts
torch.save(ra.PopularSamplerModel(zipf_counts(10_000_001, 100_000_000)), '/tmp/rsa_ps_1e7.pt')
PY
for rep in 1 2 3; do
for v in default pipeall; do
  lib=$REPO/recstudio_amd/librecstudio_amd_$v.so
  [ $v = default ] && lib=$REPO/recstudio_amd/librecstudio_amd.so
  for B in 4096 65536; do
    echo -n "$v B=$B " | tee -a gpurun_out/r3f/sweep2.log
    RSA_LIB=$lib B=$B timeout 300 python tools/exp_small.py 2>/dev/null | grep "^{" | tee -a gpurun_out/r3f/sweep2.log
  done
done
done
timeout 900 python -m pytest tests -m gpu -q -x -k "parity or round2" 2>&1 | grep -E "passed|failed"
