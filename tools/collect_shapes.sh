#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: for every bench figure that quotes a fraction of a peak, three
# rocprofv3 passes of that figure alone (tools/prof_shapes.py) -- --kernel-trace --stats, --pmc FETCH_SIZE, --pmc
# WRITE_SIZE (separate passes: the two counters do not fit one) -- summarised into
# gpurun_out/profiles_$ROUND/${ROUND}_kernel_profiles.{json,txt} (copy those to profiles/).  PASSES="trace" collects the
# kernel times only.  tools/summarize_shapes.py deletes the rocpd databases of a shape once it has read them.
export ROUND=${ROUND:-r06}
SHAPES=${SHAPES:-"headline_N1e7_popular_n64_B65536 N1e7_popular_n64_B4096 N1e7_popular_n64_B16384 N1e8_uniform_n64_B65536 N1e8_uniform_n1024_B4096 N1e8_popular_n64_B65536 ssm_N1e6_popular_n256_B8192 sharded_world1_step sharded_world1_train sgd_step_N1e7_popular_n64_B65536 adam_step_N1e7_popular_n64_B65536 train_step_N1e7_popular_n64_B65536 fullscore_lse_B2048_N1e6 fullscore_top100_B2048_N1e6 softmax_dx_B2048_N1e6 softmax_flash_fwd_B2048_N1e6 softmax_dw_B2048_N1e6 softmax_dq_nowrite_B2048_N1e6 softmax_train_B2048_N1e6 seg_gather_B8192_L50 queue_N1e7_popular_n64_B4096x16 queue_N1e7_popular_n64_B16384x4 sharded_world1_train_ssm sharded_world1_train_nondet"}
PASSES=${PASSES:-"trace fetch write"}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$ROUND
DST=$REPO/gpurun_out/profiles_$ROUND
mkdir -p $OUT $DST
export TMPDIR=/tmp
# (the counter passes need bytes, not times: no output-set selection, no placement probes -- every launch is serialised there)
for s in $SHAPES; do
  cd /tmp
  for p in $PASSES; do
    case $p in
      trace) timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/$s/trace -o p -- python $REPO/tools/prof_shapes.py $s 100 > $OUT/$s.trace.log 2>&1 ;;
      fetch) timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/$s/fetch -o p -- env PROF_WARM_MS=0 PROF_OUT_SETS=1 RSA_PLACEMENT=0 python $REPO/tools/prof_shapes.py $s 8 > $OUT/$s.fetch.log 2>&1 ;;
      write) timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/$s/write -o p -- env PROF_WARM_MS=0 PROF_OUT_SETS=1 RSA_PLACEMENT=0 python $REPO/tools/prof_shapes.py $s 8 > $OUT/$s.write.log 2>&1 ;;
    esac
  done
  cd $REPO
  python tools/summarize_shapes.py $OUT $DST $s >> $OUT/summary.log 2>&1
done
tail -${TAIL:-80} $OUT/summary.log
