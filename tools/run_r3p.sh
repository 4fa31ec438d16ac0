#!/bin/bash
REPO=$(pwd)
lib() { for v in "$@"; do echo -n " $v=$REPO/recstudio_amd/librecstudio_amd_$v.so"; done; }
for s in dq item; do SHAPE=$s python tools/exp_sorted_ab.py $(lib u2 u4 u6 u16 u8w5) 2>&1 | grep "^{\|Error\|error" | head -3; done
