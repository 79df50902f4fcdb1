#!/bin/bash
set -o pipefail
O=gpurun_out/r03i
mkdir -p $O
timeout -k 10 900 python tools/walk_diag.py --big 65536 > $O/walk_diag.jsonl 2> $O/walk_diag.err || { tail -20 $O/walk_diag.err; exit 1; }
cut -c1-420 $O/walk_diag.jsonl
