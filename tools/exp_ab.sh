#!/bin/bash
# GPU box: A/B a -D switch of the fused forward on the bench headline.  usage: exp_ab.sh MACRO v1 v2 ...
M=$1; shift
for v in "$@"; do
  rm -f recstudio_amd/csrc/rsa_fused.o
  make -C recstudio_amd/csrc -j8 CXXFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -ffp-contract=off -D$M=$v" > /dev/null 2>&1
  for rep in 1 2; do
    echo "== $M=$v: $(python bench.py --no-cpu-baseline --no-sweep --steps 100 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['roofline']['avg_kernel_ms'], j['roofline']['frac'], j['value'])")"
  done
done
rm -f recstudio_amd/csrc/rsa_fused.o
make -C recstudio_amd/csrc -j8 > /dev/null 2>&1
