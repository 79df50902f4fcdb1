mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_dbg -- python $GRAFT_REPO_ROOT/tools/dbg_mfma.py > $GRAFT_REPO_ROOT/gpurun_out/dbg_mfma.log 2>&1
cd $GRAFT_REPO_ROOT; grep -v "^W2026\|^E2026\|^I2026" gpurun_out/dbg_mfma.log | tail -8; cat gpurun_out/prof_dbg/*/*kernel_stats.csv | cut -c1-160 | head -8
