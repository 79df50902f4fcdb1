#!/bin/bash
# PMC passes over the kinematics kernel alone (separate runs, --kernel-trace only) -> gpurun_out/pmc_kin/TAG_p*
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
TAG=${1:-kin}
O=$R/gpurun_out/pmc_kin; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 60 rocprofv3 -L > $O/counters.txt 2>&1
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR"
P3="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum TCC_WRITE_sum"
P4="TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCC_REQ_sum TCC_WRITEBACK_sum TCC_TAG_STALL_sum"
n=1
for P in "$P1" "$P2" "$P3" "$P4"; do
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/${TAG}_p$n -- python $R/bench.py --workload kin --batch 65536 --steps 10 --warmup 3 > $O/${TAG}_p$n.log 2>&1 || { tail -5 $O/${TAG}_p$n.log; }
  n=$((n+1))
done
python $R/tools/pmc/summarize_ik.py $O kin_jacobians > $O/${TAG}_summary.json
cat $O/${TAG}_summary.json
