mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
timeout 400 python tools/exp_r2.py pop7 pop8 ssm small > gpurun_out/r2a/exp_default.log 2>&1
RSA_LIB=$PWD/recstudio_amd/librecstudio_amd_qgpipe.so timeout 200 python tools/exp_r2.py pop7 > gpurun_out/r2a/exp_qgpipe.log 2>&1
RSA_LIB=$PWD/recstudio_amd/librecstudio_amd_ssm8.so timeout 200 python tools/exp_r2.py ssm > gpurun_out/r2a/exp_ssm8.log 2>&1
RSA_LIB=$PWD/recstudio_amd/librecstudio_amd_noahead.so timeout 300 python tools/exp_r2.py pop7 pop8 > gpurun_out/r2a/exp_noahead.log 2>&1
grep -h "RESULT" gpurun_out/r2a/exp_*.log | cut -c1-3000
