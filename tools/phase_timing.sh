#!/bin/bash
# Diagnostic only: builds libwcqp variants whose IK kernel returns before phase N and times
# them, to see where the IK kernel spends its time.  Never part of the product build.
set -e
cd "$(dirname "$0")/../walking-controllers_amd/csrc"
mkdir -p build/diag
for n in 1 2 3 4 5 6; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -DWCQP_IK_PHASE_STOP=$n -x hip -c ik.hip -o build/diag/ik_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/diag/libwcqp_stop$n.so build/common.cpp.o build/mpc.hip.o build/diag/ik_$n.o
done
