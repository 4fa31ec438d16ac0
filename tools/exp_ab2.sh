#!/bin/bash
# GPU box: A/B a -D switch of one source file on the bench train step.  usage: exp_ab2.sh FILE MACRO v1 v2 ...
F=$1; M=$2; shift; shift
for v in "$@"; do
  rm -f recstudio_amd/csrc/$F.o
  make -C recstudio_amd/csrc -j8 CXXFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -ffp-contract=off -D$M=$v" > /dev/null 2>&1
  echo "== $M=$v: $(python bench.py --no-cpu-baseline --no-sweep --steps 100 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['train_step']['ms_per_step'], j['train_step']['forward_ms'], j['train_step']['two_pass_ms_per_step'], j['roofline']['avg_kernel_ms'])")"
done
rm -f recstudio_amd/csrc/$F.o
make -C recstudio_amd/csrc -j8 > /dev/null 2>&1
