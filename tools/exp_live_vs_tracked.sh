cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_shard.py -q -x -k "determin or staged or overflow" > gpurun_out/det_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/det_tests.log | tail -2
for s in queue_N1e7_popular_n64_B4096x16 sharded_world1_step headline_N1e7_popular_n64_B65536; do
  for i in 1 2; do python tools/prof_shapes.py $s 100 2>/dev/null | grep event_us | sed "s/^/plain$i /"; done
  rocprofv3 --kernel-trace --stats -d /tmp/lv_$s -- python tools/prof_shapes.py $s 100 2>/dev/null | grep event_us | sed "s/^/rocprof /"
  python tools/prof_shapes.py $s 100 2>/dev/null | grep event_us | sed "s/^/plain3 /"
done
