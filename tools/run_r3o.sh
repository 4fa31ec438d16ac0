#!/bin/bash
REPO=$(pwd)
lib() { for v in "$@"; do echo -n " $v=$REPO/recstudio_amd/librecstudio_amd_$v.so"; done; }
SHAPE=ssm_train python tools/exp_fwd_ab.py $(lib ssm3 ssmb2) 2>&1 | grep "^{"
SHAPE=walk_train python tools/exp_fwd_ab.py $(lib walk1) 2>&1 | grep "^{"
SHAPE=headline_pop python tools/exp_fwd_ab.py $(lib b4) 2>&1 | grep "^{"
python -m pytest tests/test_gpu_round2.py tests/test_gpu_shard.py tests/test_gpu_parity.py -m gpu -q -x --deselect tests/test_gpu_shard.py::test_two_gpus_rccl 2>&1 | tail -2
