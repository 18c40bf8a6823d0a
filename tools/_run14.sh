mkdir -p gpurun_out/r2n
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r2n/pmc_wgrad9 -o wg -- python3 $R/tools/kbench.py wgrad6430 5 > $R/gpurun_out/r2n/pmc1.log 2>&1; echo "pmc1 rc=$?"
cd $R && find gpurun_out/r2n/pmc_wgrad9 -name "*.csv" | head; tail -3 gpurun_out/r2n/pmc1.log
