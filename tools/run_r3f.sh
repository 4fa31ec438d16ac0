#!/bin/bash
# headline-size A/B of compile-time variants, alternating processes on one box
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
for v in default b4 pin1 pin2 cap1k cap4k ntst0 qnt; do
  lib=$REPO/recstudio_amd/librecstudio_amd_$v.so
  [ $v = default ] && lib=$REPO/recstudio_amd/librecstudio_amd.so
  echo -n "$v B=65536 " | tee -a gpurun_out/r3f/sweep3.log
  RSA_LIB=$lib B=65536 timeout 300 python tools/exp_small.py 2>/dev/null | grep "^{" | tee -a gpurun_out/r3f/sweep3.log
done
done
