#!/bin/bash
# GPU suite incl. the device hull builder test
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/run15_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/run15_tests.log
tail -5 gpurun_out/run15_tests.log
