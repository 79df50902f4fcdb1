#!/bin/bash
set -o pipefail
D=$PWD/walking-controllers_amd/csrc/build/diag
bash tools/pmc/collect_ik.sh w2_mfma 3 && bash tools/pmc/collect_ik.sh w2_valu 2 && bash tools/pmc/collect_ik.sh w3_valu 2 $D/libwcqp_w3.so
python tools/pmc/summarize_ik.py gpurun_out/pmc_ik > gpurun_out/pmc_ik_summary.json; tail -5 gpurun_out/pmc_ik_summary.json
