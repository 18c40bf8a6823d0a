#!/bin/bash
# instruction mix of the two persistent launches (PMC, per dispatch averages): how many VALU / SALU / LDS / MFMA instructions a launch issues and how many
# cycles each class is active — the evidence behind "the short phases are issue-bound" (DESIGN.md lessons 28, 37, 39).  usage: gpu_resident_mix.sh <commit>
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/resmix; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-parity"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_BUSY_CYCLES --output-format csv -d $O/a -o a -- $CMD > $O/a.log 2>&1 || { tail -5 $O/a.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/b -o b -- $CMD > $O/b.log 2>&1 || { tail -5 $O/b.log; exit 1; }
python3 - "$O" "${1:-?}" <<'PY'
import csv, glob, json, sys, collections
O, commit = sys.argv[1], sys.argv[2]
out = {'round': 'r03', 'commit': commit, 'command': 'rocprofv3 --kernel-trace --pmc <SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_BUSY_CYCLES | SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES> -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-parity (two passes); values: average per dispatch, summed over the device', 'kernels': {}}
for d in ('a', 'b'):
    for f in glob.glob(f'{O}/{d}/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name']
            for name in ('resident_fwd_kernel', 'resident_bwd_kernel'):
                if name in k: acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
        for name, cs in acc.items():
            for c, v in cs.items(): out['kernels'].setdefault(name, {})[c] = round(sum(v) / len(v), 1)
json.dump(out, open(f'{O}/r03_pmc_resident_mix.json', 'w'), indent=1)
for k, v in out['kernels'].items(): print(k, v)
PY
rm -rf $O/a $O/b
