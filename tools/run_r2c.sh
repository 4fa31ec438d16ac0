mkdir -p gpurun_out/r2c
for v in "" _pin1b4 _pin2b4 _pin1b8 _pin2b8; do
  RSA_LIB=$PWD/recstudio_amd/librecstudio_amd$v.so timeout 300 python tools/exp_r2.py pop7 pop8 small > gpurun_out/r2c/exp$v.log 2>&1
done
grep -h RESULT gpurun_out/r2c/*.log | cut -c1-2500
