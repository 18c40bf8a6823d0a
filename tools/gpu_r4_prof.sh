#!/bin/bash
# kernel statistics of one wide spec: usage gpu_r4_prof.sh <spec> <N> <mode: eager,bf16 | graph,bf16 | ...>; result gpurun_out/r4/<spec>_<N>_kernel_stats.csv
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
spec=$1; n=$2; arg=$3
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o w -- python3 $R/tools/wide_bench.py $spec $n 10 $arg > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
cp $(find $O/st -name "*kernel_stats.csv" | head -1) $O/${spec}_${n}_kernel_stats.csv
rm -rf $O/st
tail -1 $O/prof.log
