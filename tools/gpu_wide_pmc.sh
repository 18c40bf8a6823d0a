#!/bin/bash
# PMC record of the bf16 STORAGE path of cifar100_deep (bs 512): HBM GB/s of the streaming normalisation / pooling kernels, matrix-pipe utilisation of the
# bf16 convolution kernels.  usage: gpu_wide_pmc.sh <commit>
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/widepmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/wide_bench.py cifar100_deep_model 512 6 eager,bf16"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- $CMD > $O/f.log 2>&1 || { tail -5 $O/f.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- $CMD > $O/w.log 2>&1 || { tail -5 $O/w.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/m -o m -- $CMD > $O/m.log 2>&1 || { tail -5 $O/m.log; exit 1; }
python3 $R/tools/pmc_kernels.py $O/r03_pmc_cifar100_deep_bf16.json r03 ${1:-?} "rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES> -- python3 tools/wide_bench.py cifar100_deep_model 512 6 eager,bf16 (separate passes)" \
  $O/f $O/w $O/m nc_apply_kernel nc_bwd_apply_kernel nc_sample_sums_kernel pool_fwd_kernel pool_bwd_kernel conv_bf16_halo_kernel conv_bf16_kernel conv_bf16_wgrad9_kernel im2col_bf16_kernel > $O/summary.txt || { tail -5 $O/summary.txt; exit 1; }
python3 -c "import json; r=json.load(open('$O/r03_pmc_cifar100_deep_bf16.json')); [print(k, v) for k, v in r['kernels'].items()]"
rm -rf $O/f $O/w $O/m
# the fp32 MFMA kernels of the same model (matrix-pipe utilisation, traffic)
CMD="python3 $R/tools/wide_bench.py cifar100_deep_model 512 4 eager"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- $CMD > $O/f.log 2>&1 || { tail -5 $O/f.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- $CMD > $O/w.log 2>&1 || { tail -5 $O/w.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/m -o m -- $CMD > $O/m.log 2>&1 || { tail -5 $O/m.log; exit 1; }
python3 $R/tools/pmc_kernels.py $O/r03_pmc_mfma_utilisation.json r03 ${1:-?} "rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES> -- python3 tools/wide_bench.py cifar100_deep_model 512 4 eager (fp32, separate passes)" \
  $O/f $O/w $O/m conv_mfma_win_kernel conv_mfma_wgrad9_kernel conv_mfma_kernel > $O/summary2.txt || { tail -5 $O/summary2.txt; exit 1; }
python3 -c "import json; r=json.load(open('$O/r03_pmc_mfma_utilisation.json')); [print(k, v) for k, v in r['kernels'].items()]"
rm -rf $O/f $O/w $O/m
