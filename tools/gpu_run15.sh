mkdir -p gpurun_out
L=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_stamps.so
WCQP_LIB_PATH=$L timeout -k 10 100 python tools/stamps.py 4096 0.5 > gpurun_out/stamps.log 2>&1
WCQP_LIB_PATH=$L timeout -k 10 100 python tools/stamps.py 65536 0.5 >> gpurun_out/stamps.log 2>&1
WCQP_LIB_PATH=$L timeout -k 10 100 python tools/stamps.py 64 100 >> gpurun_out/stamps.log 2>&1
grep median gpurun_out/stamps.log
