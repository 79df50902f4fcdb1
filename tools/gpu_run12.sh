mkdir -p gpurun_out; : > gpurun_out/time_alg.log
for L in "" $PWD/walking-controllers_amd/csrc/build/diag/libwcqp_w2.so; do
 for B in 4096 65536; do for V in 100 0.5; do
  WCQP_LIB_PATH=$L timeout -k 10 200 python tools/time_alg.py $B $V >> gpurun_out/time_alg.log 2>&1
 done; done; done
grep ms gpurun_out/time_alg.log
