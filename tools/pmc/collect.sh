#!/bin/bash
# PMC passes (separate runs, --kernel-trace only; never combined with other trace domains).
# Usage on the GPU box: bash tools/pmc/collect.sh   -> gpurun_out/pmc/*
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/calib_$C -- $R/tools/_build/pmc_calib > $O/calib_$C.log 2>&1
  for B in 4096 65536; do
    timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/bench_${B}_$C -- python $R/bench.py --steps 20 --warmup 5 --batch $B --no-cpu-baseline > $O/bench_${B}_$C.log 2>&1
  done
done
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_F64 --kernel-trace --output-format csv -d $O/bench_65536_SQ -- python $R/bench.py --steps 20 --warmup 5 --batch 65536 --no-cpu-baseline > $O/bench_65536_SQ.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/bench_65536_LDS -- python $R/bench.py --steps 20 --warmup 5 --batch 65536 --no-cpu-baseline > $O/bench_65536_LDS.log 2>&1
python $R/tools/pmc/summarize.py $O > $O/summary.txt 2>&1; cat $O/summary.txt
