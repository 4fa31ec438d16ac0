#!/bin/bash
REPO=$(pwd)
lib() { for v in "$@"; do echo -n " $v=$REPO/recstudio_amd/librecstudio_amd_$v.so"; done; }
for s in walk walk_1e8; do SHAPE=$s python tools/exp_fwd_ab.py $(lib w5 wb2 wb8) 2>&1 | grep "^{"; done
for s in b16384 b4096; do SHAPE=$s python tools/exp_fwd_ab.py $(lib pb4w5) 2>&1 | grep "^{"; done
SHAPE=ssm python tools/exp_fwd_ab.py $(lib sb2 sb8) 2>&1 | grep "^{"
