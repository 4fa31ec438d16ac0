#!/bin/bash
# Build another copy of the library with extra -D switches for A/B measurements on the GPU box:
#   tools/build_variant.sh qg4 -DRSA_QG_BATCH=4
#   VARIANT_FILES=rsa_fullscore tools/build_variant.sh dq1 -DRSA_FS_DQ_MIN_BLOCKS=1
# writes recstudio_amd/librecstudio_amd_qg4.so (git-ignored, travels with gpurun); select it with
#   RSA_LIB=$PWD/recstudio_amd/librecstudio_amd_qg4.so python bench.py ...
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/recstudio_amd/csrc/variant_$NAME
mkdir -p $OUT
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -ffp-contract=off $*"
for f in rsa_misc rsa_sample rsa_fused rsa_loss rsa_backward rsa_fullscore rsa_shard rsa_sorted rsa_owner rsa_step rsa_dx; do
  # only the sources named in VARIANT_FILES (default: the fused forward) are recompiled with the switches; the other
  # objects are reused from the default build
  if [[ " ${VARIANT_FILES:-rsa_fused} " == *" $f "* ]] || [ ! -f $ROOT/recstudio_amd/csrc/$f.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -c $ROOT/recstudio_amd/csrc/$f.hip -o $OUT/$f.o &
  else
    cp $ROOT/recstudio_amd/csrc/$f.o $OUT/$f.o
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/*.o -o $ROOT/recstudio_amd/librecstudio_amd_$NAME.so
rm -rf $OUT
echo built $ROOT/recstudio_amd/librecstudio_amd_$NAME.so
