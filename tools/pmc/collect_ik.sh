#!/bin/bash
# PMC passes over the IK kernel alone (separate runs, --kernel-trace only).
#   bash tools/pmc/collect_ik.sh TAG ALG [LIB]  -> gpurun_out/pmc_ik/TAG_*
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
TAG=$1; ALG=$2; [ -n "$3" ] && export WCQP_LIB_PATH=$3
O=$R/gpurun_out/pmc_ik; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
P2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY"
P3="SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_IFETCH SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"
n=1
for P in "$P1" "$P2" "$P3"; do
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/${TAG}_p$n -- python $R/tools/pmc/ik_only.py 65536 $ALG > $O/${TAG}_p$n.log 2>&1 || { tail -5 $O/${TAG}_p$n.log; }
  n=$((n+1))
done
