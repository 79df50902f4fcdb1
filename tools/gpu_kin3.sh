#!/bin/bash
set -o pipefail
L=gpurun_out/kin.log
mkdir -p gpurun_out
WCQP_LIB_PATH=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_kp2.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_tick_pipeline.py -m gpu -q -x -k "kin or hull or tick" > $L 2>&1 || { tail -40 $L; exit 1; }
tail -2 $L
bash tools/gpu_kin2.sh
