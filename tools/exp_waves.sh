#!/bin/bash
# GPU box: A/B the fused forward at different register caps (launch_bounds min waves/SIMD).
for w in 1 5 6 8; do
  rm -f recstudio_amd/csrc/rsa_fused.o
  make -C recstudio_amd/csrc -j8 CXXFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -ffp-contract=off -DRSA_FWD_MIN_WAVES=$w" > /dev/null 2>&1
  echo "== min waves/SIMD = $w"
  python tools/exp_fwd.py 2>&1 | grep -E "given ids|uniform sampler|guide_log2=24 logp=True pairs=True"
done
rm -f recstudio_amd/csrc/rsa_fused.o
make -C recstudio_amd/csrc -j8 > /dev/null 2>&1
