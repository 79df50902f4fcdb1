#!/bin/bash
# Diagnostic only: builds libwcqp variants that differ in the flags given to the IK kernels
#   tools/build_variant.sh NAME [-Dflag ...]   ->  walking-controllers_amd/csrc/build/diag/libwcqp_NAME.so
set -e
name=$1; shift
cd "$(dirname "$0")/../walking-controllers_amd/csrc"
make -s >/dev/null      # NOTE: rebuilds the PRODUCT library from the working tree as well - A/B a source change against a variant built from a stash, not against "the product"
mkdir -p build/diag
for f in ik ik2 ik3 ik4 kin tick; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -DWCQP_DIAG_KERNELS "$@" -x hip -c $f.hip -o build/diag/${f}_$name.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/diag/libwcqp_$name.so build/common.cpp.o build/mpc.hip.o build/diag/ik_$name.o build/diag/ik2_$name.o build/diag/ik3_$name.o build/diag/ik4_$name.o build/diag/tick_$name.o build/hull.hip.o build/diag/kin_$name.o build/host_WalkingControllers.o
echo built build/diag/libwcqp_$name.so
