#!/bin/bash
# kernel statistics of the bench workload replayed as a hipGraph (DCV_REPLAY_LIST=0: the fallback of the launch-list replay)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DCV_REPLAY_LIST=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --steps 50 --warmup 10 --no-secondary --no-cpu-baseline --no-parity > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
head -8 $O/kernel_stats.csv | cut -c1-200
tail -2 $O/stats.log | cut -c1-300
rm -rf $O/stats
