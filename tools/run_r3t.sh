#!/bin/bash
REPO=$(pwd)
lib() { for v in "$@"; do echo -n " $v=$REPO/recstudio_amd/librecstudio_amd_$v.so"; done; }
for s in headline_pop headline_uni given train n1e8_uni n1e8_pop; do SHAPE=$s python tools/exp_fwd_ab.py $(lib b4c1k b4p1c1k b4c768 b4c1280 c1k b4c512) 2>&1 | grep "^{"; done
