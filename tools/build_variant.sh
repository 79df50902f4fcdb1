#!/bin/bash
# Diagnostic only: builds libwcqp variants that differ in the flags given to the kernels
#   tools/build_variant.sh NAME [-Dflag ...]   ->  walking-controllers_amd/csrc/build/diag/libwcqp_NAME.so   (pick it up with WCQP_LIB_PATH)
# WCQP_VARIANT_NO_DIAG=1: without -DWCQP_DIAG_KERNELS (the product's own flags + the ones given)
set -e
name=$1; shift
cd "$(dirname "$0")/../walking-controllers_amd/csrc"
make -s >/dev/null      # NOTE: rebuilds the PRODUCT library from the working tree as well - A/B a source change against a variant built from a stash, not against "the product"
mkdir -p build/diag
diag=-DWCQP_DIAG_KERNELS
[ -n "$WCQP_VARIANT_NO_DIAG" ] && diag=
pids=()
for f in mpc ik ik2 ik3 ik4 kin tick; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. $diag "$@" -x hip -c $f.hip -o build/diag/${f}_$name.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/diag/libwcqp_$name.so build/common.cpp.o build/diag/mpc_$name.o build/diag/ik_$name.o build/diag/ik2_$name.o build/diag/ik3_$name.o build/diag/ik4_$name.o build/diag/tick_$name.o build/hull.hip.o build/diag/kin_$name.o build/host_WalkingControllers.o
echo built build/diag/libwcqp_$name.so
