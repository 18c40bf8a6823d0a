#!/bin/bash
# which kernels of a CAPTURED wide step are not this library's (they keep the step from replaying as a launch list)?
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in cifar100_deep_model resnet18_style_model; do
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fk -o s -- python3 $R/tools/wide_bench.py $m 512 20 graph,bf16 > $O/foreign.log 2>&1 || { tail -5 $O/foreign.log; exit 1; }
grep "img/s" $O/foreign.log | tail -1
python3 - <<PY
import csv, glob
f = glob.glob('$O/fk/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'anonymous namespace' in n or n.startswith('dcv') or 'im2col_bf16' in n: continue
    print(f"  {n[:150]}  calls {r['Calls']}  total {int(r['TotalDurationNs']) / 1e3:.1f} us")
PY
rm -rf $O/fk
done
