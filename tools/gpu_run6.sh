mkdir -p gpurun_out
D=walking-controllers_amd/csrc/build/diag
: > gpurun_out/phase_timing.log
for n in 1 2 3 4 5 6; do WCQP_LIB_PATH=$PWD/$D/libwcqp_stop$n.so timeout -k 10 120 python tools/phase_timing.py 65536 0.5 >> gpurun_out/phase_timing.log 2>&1; done
timeout -k 10 120 python tools/phase_timing.py 65536 0.5 >> gpurun_out/phase_timing.log 2>&1
timeout -k 10 120 python tools/phase_timing.py 65536 100.0 >> gpurun_out/phase_timing.log 2>&1
timeout -k 10 120 python tools/phase_timing.py 65536 0.3 >> gpurun_out/phase_timing.log 2>&1
cat gpurun_out/phase_timing.log
