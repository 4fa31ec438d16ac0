#!/bin/bash
# in-process A/B of the segment-form scoring kernel: default (4-row batches, 123 VGPRs) vs variants
REPO=$(pwd)
L=""
for v in b8s1 b4s5 b2s1 b2s5 b4p1 b4p2; do L="$L $v=$REPO/recstudio_amd/librecstudio_amd_$v.so"; done
python tools/exp_seg.py $L 2>&1 | grep "^{"
WORLD=8 python tools/exp_seg.py $L 2>&1 | grep "^{"
NEG=64 B=65536 python tools/exp_seg.py $L 2>&1 | grep "^{"
