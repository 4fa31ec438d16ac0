#!/bin/bash
# A/B of the segment-form scoring kernel's occupancy: default (4 waves/SIMD, 24 B of scratch) vs seg1 (129 VGPRs, 3 waves)
# vs b4s1 (4-row batches, 123 VGPRs, 4 waves); alternating processes, sharded forward step at world size 1
REPO=$(pwd)
for rep in 1 2 3; do
  for v in default seg1 b4s1; do
    if [ $v = default ]; then unset RSA_LIB; else export RSA_LIB=$REPO/recstudio_amd/librecstudio_amd_$v.so; fi
    echo "$v $(STEP=fwd timeout 300 python tools/exp_shard2.py 2>&1 | grep '^{' | cut -c1-200)"
    echo "$v n64 $(STEP=fwd NEG=64 B=65536 timeout 300 python tools/exp_shard2.py 2>&1 | grep '^{' | cut -c1-200)"
  done
done
