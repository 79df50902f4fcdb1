#!/bin/bash
# PMC passes over the default bench command (one-launch steps, three batches in flight) -> gpurun_out/pmc_pair/TAG_p*
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
TAG=${1:-pair}; B=${2:-4096}
O=$R/gpurun_out/pmc_pair; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_F64"
n=1
for P in "$P1" "$P2"; do
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/${TAG}_p$n -- python3 $R/bench.py --batch $B --steps 60 --warmup 5 --no-cpu-baseline > $O/${TAG}_p$n.log 2>&1 || { tail -5 $O/${TAG}_p$n.log; exit 1; }
  n=$((n+1))
done
python3 $R/tools/pmc/summarize_ik.py $O qp_pair_kernel > $O/${TAG}_summary.json
cat $O/${TAG}_summary.json
