#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2p
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2p/bench -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-secondary --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2p/bench.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/r2p/bench.log; exit 1; }
find $GRAFT_REPO_ROOT/gpurun_out/r2p/bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/gpurun_out/r2p/kernel_stats.csv
cut -c1-150 $GRAFT_REPO_ROOT/gpurun_out/r2p/kernel_stats.csv | head -30
