#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2u
cd /tmp && export TMPDIR=/tmp
for g in 1 2; do
export DCV_BF16_WGRAD9_GROUPS=$g
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2u/g$g -o b -- python3 $GRAFT_REPO_ROOT/tools/bf16_bench.py 10 64,32,64 256,16,256 512,8,512 > $GRAFT_REPO_ROOT/gpurun_out/r2u/g$g.log 2>&1 || exit 1
done
