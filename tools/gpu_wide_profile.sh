#!/bin/bash
# kernel statistics of the wide specs (BASELINE configs 3 / 5): usage gpu_wide_profile.sh <tag> [mode...]   modes: bf16 | fp32
# results: gpurun_out/wide/<spec>_<mode>_kernel_stats.csv + a top-of-list print
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wide; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in ${@:2}; do
  for job in "cifar100_deep_model 512" "resnet18_style_model 64"; do
    set -- $job; spec=$1; n=$2
    arg=$([ $mode == bf16 ] && echo "eager,bf16" || echo "eager")
    timeout -k 10 300 python3 $R/tools/wide_bench.py $spec $n 10 $arg > $O/${spec}_${mode}.txt 2>&1 || { tail -5 $O/${spec}_${mode}.txt; exit 1; }
    tail -1 $O/${spec}_${mode}.txt
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o w -- python3 $R/tools/wide_bench.py $spec $n 10 $arg > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
    cp $(find $O/st -name "*kernel_stats.csv" | head -1) $O/${spec}_${mode}_kernel_stats.csv
    rm -rf $O/st
    python3 - <<PY
import csv
rows = list(csv.DictReader(open('$O/${spec}_${mode}_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'--- $spec $mode: {tot / 14 / 1e6:.2f} ms of kernels per step (14 steps profiled)')
for r in rows[:16]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:9.1f} us {float(r['Percentage']):6.2f} %")
PY
  done
done
