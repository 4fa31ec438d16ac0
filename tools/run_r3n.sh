#!/bin/bash
# in-process A/B (tools/exp_fwd_ab.py) of compile-time variants of the fused forward kernels, shape by shape
REPO=$(pwd)
lib() { for v in "$@"; do echo -n " $v=$REPO/recstudio_amd/librecstudio_amd_$v.so"; done; }
for s in headline_pop headline_uni given; do SHAPE=$s python tools/exp_fwd_ab.py $(lib b4 b16 pin1 pin2 mw5 cap1k cap4k pipeall) 2>&1 | grep "^{"; done
for s in b4096 b16384; do SHAPE=$s python tools/exp_fwd_ab.py $(lib pb4 pb16 cap1k) 2>&1 | grep "^{"; done
SHAPE=train python tools/exp_fwd_ab.py $(lib qg8 qgmw4 cap1k cap4k) 2>&1 | grep "^{"
SHAPE=ssm python tools/exp_fwd_ab.py $(lib ssm8 ssmw4 qg8) 2>&1 | grep "^{"
SHAPE=walk python tools/exp_fwd_ab.py $(lib walkT qg8) 2>&1 | grep "^{"
SHAPE=upd python tools/exp_fwd_ab.py $(lib upd3 qg8) 2>&1 | grep "^{"
