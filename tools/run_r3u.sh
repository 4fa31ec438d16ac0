#!/bin/bash
REPO=$(pwd)
lib() { for v in "$@"; do echo -n " $v=$REPO/recstudio_amd/librecstudio_amd_$v.so"; done; }
for s in upd b16384 headline_pop; do SHAPE=$s python tools/exp_fwd_ab.py $(lib b4p1c1k c1k) 2>&1 | grep "^{"; done
python tools/exp_seg.py $(lib b4p1c1k c1k) 2>&1 | grep "^{"
WORLD=8 python tools/exp_seg.py $(lib b4p1c1k c1k) 2>&1 | grep "^{"
NEG=64 B=65536 python tools/exp_seg.py $(lib b4p1c1k c1k) 2>&1 | grep "^{"
