#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_stamps.so
: > gpurun_out/stamps.log
for args in "64 100 4" "65536 0.5 4" "64 100 3"; do
  WCQP_LIB_PATH=$L timeout -k 10 100 python tools/stamps.py $args >> gpurun_out/stamps.log 2>&1 || { tail -20 gpurun_out/stamps.log; exit 1; }
done
grep median gpurun_out/stamps.log
bash tools/pmc/collect_ik.sh ik3 4 > /dev/null 2>&1
python tools/pmc/summarize_ik.py gpurun_out/pmc_ik > gpurun_out/pmc_ik_summary.json
