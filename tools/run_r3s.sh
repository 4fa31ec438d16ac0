#!/bin/bash
REPO=$(pwd)
for s in headline_uni headline_pop given; do SHAPE=$s python tools/exp_fwd_ab.py phx=$REPO/recstudio_amd/librecstudio_amd_phx.so 2>&1 | grep "^{"; done
