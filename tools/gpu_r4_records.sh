#!/bin/bash
# round-4 measurement records in one gpurun call (usage: gpurun -- "bash tools/gpu_r4_records.sh <commit>"); results under gpurun_out/r4rec/, copied to profiles/r04_* by hand
#   bench line; kernel statistics of the bench command; PMC traffic of the persistent launches; kernel statistics of both wide specs (bf16); PMC (HBM bytes, matrix-pipe
#   busy) of the bf16 path of cifar100_deep and of the bf16 convolution entry points at the bench shapes
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4rec; C=${1:-?}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 $R/bench.py > $O/r04_bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cut -c1-400 $O/r04_bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --steps 50 --warmup 10 --no-secondary --no-cpu-baseline --no-parity > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/r04_bench_steps50_kernel_stats.csv; rm -rf $O/stats
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-parity > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-parity > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
python3 $R/tools/pmc_traffic.py $(find $O/fetch -name "*counter_collection.csv" | head -1) $(find $O/write -name "*counter_collection.csv" | head -1) $O/r04_pmc_resident_traffic.json r04 $C > /dev/null
rm -rf $O/fetch $O/write
echo "--- wide specs"
for job in "cifar100_deep_model 512" "resnet18_style_model 512" "resnet18_style_model 64"; do
  set -- $job
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o w -- python3 $R/tools/wide_bench.py $1 $2 10 eager,bf16 > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
  cp $(find $O/st -name "*kernel_stats.csv" | head -1) $O/r04_${1%_model}_bs$2_bf16_kernel_stats.csv; rm -rf $O/st
  grep "img/s" $O/prof.log | tail -1
  python3 $R/tools/wide_bench.py $1 $2 10 graph,bf16 2>&1 | grep "img/s" | tail -1
done
echo "--- PMC, cifar100_deep bf16"
CMD="python3 $R/tools/wide_bench.py cifar100_deep_model 512 6 eager,bf16"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- $CMD > $O/f.log 2>&1 || { tail -5 $O/f.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- $CMD > $O/w.log 2>&1 || { tail -5 $O/w.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/m -o m -- $CMD > $O/m.log 2>&1 || { tail -5 $O/m.log; exit 1; }
python3 $R/tools/pmc_kernels.py $O/r04_pmc_cifar100_deep_bf16.json r04 $C "rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES> -- python3 tools/wide_bench.py cifar100_deep_model 512 6 eager,bf16 (separate passes)" \
  $O/f $O/w $O/m nc_apply_kernel nc_bwd_apply_kernel nc_sample_sums_kernel conv_bf16_halo_kernel conv_bf16_halo_rf_kernel conv_bf16_kernel conv_bf16_rfk_kernel conv_bf16_wgrad9p_kernel conv_bf16_wgradg_kernel im2col_bf16_kernel prep_bf16_tile_kernel mfma_wgrad9_reduce > $O/summary.txt || { tail -5 $O/summary.txt; exit 1; }
python3 -c "import json; r=json.load(open('$O/r04_pmc_cifar100_deep_bf16.json')); [print(k, v.get('avg_duration_us'), v.get('hbm_frac_of_8TBps'), v.get('mfma_busy_frac_of_simd_cycles')) for k, v in r['kernels'].items()]"
rm -rf $O/f $O/w $O/m
echo "--- PMC, bf16 entry points at the bench shapes"
for case in 256,16,256 512,8,512; do
  CMD="python3 $R/tools/bf16_bench.py 10 $case"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- $CMD > $O/f.log 2>&1 || { tail -5 $O/f.log; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- $CMD > $O/w.log 2>&1 || { tail -5 $O/w.log; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/m -o m -- $CMD > $O/m.log 2>&1 || { tail -5 $O/m.log; exit 1; }
  python3 $R/tools/pmc_kernels.py $O/r04_pmc_bf16_${case//,/_}.json r04 $C "rocprofv3 --kernel-trace --pmc <...> -- python3 tools/bf16_bench.py 10 $case (N = 512, 3x3; separate passes)" \
    $O/f $O/w $O/m conv_bf16_halo_kernel "conv_bf16_halo_kernel<2, true, false, false>" conv_bf16_wgrad9p_kernel mfma_wgrad9_reduce > $O/summary.txt || { tail -5 $O/summary.txt; exit 1; }
  python3 -c "import json; r=json.load(open('$O/r04_pmc_bf16_${case//,/_}.json')); [print('$case', k, v.get('avg_duration_us'), v.get('mfma_busy_frac_of_simd_cycles')) for k, v in r['kernels'].items()]"
  rm -rf $O/f $O/w $O/m
done
