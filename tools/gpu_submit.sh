#!/bin/bash
# Submits ONE gpurun call and waits for a free slot: retries ONLY while gpurun answers 3 (no box / slot free: nothing ran, nothing
# was charged).  Any other outcome - success, a failing command, a refusal - ends it.   tools/gpu_submit.sh TIMEOUT 'command'
t=$1; shift
for try in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  echo "[gpu_submit] no slot (try $try), waiting"; sleep 120
done
exit 3
