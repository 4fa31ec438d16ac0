#!/bin/bash
REPO=$(pwd)
ADAM=1 SHAPE=all python tools/exp_sorted_ab.py u8=$REPO/recstudio_amd/librecstudio_amd_u8.so u6=$REPO/recstudio_amd/librecstudio_amd_u6.so 2>&1 | grep "^{\|rror" | head -3
python tools/exp_host.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | head -70
