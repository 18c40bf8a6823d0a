#!/bin/bash
# PMC counters of the bf16 window kernel on one shape: LDS activity / bank conflicts, matrix-pipe busy, waits, L2 hits (separate passes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4/pmc_halo; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CASE="${CASE:-512,8,512}"
CMD="python3 $R/tools/bf16_bench.py 5 $CASE"
rocprofv3 -L > $O/counters.txt 2>&1
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- $CMD > $O/p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $O/p$i.log; continue; }
  python3 - "$O/p$i" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not f: print('no csv'); sys.exit(0)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name']
    if 'conv_bf16' in k:
        acc[(k.split('(')[0][-60:], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(acc.items()):
    print(f'{k:62s} {c:28s} n={len(v):3d} avg={sum(v)/len(v):.4g}')
PY
  rm -rf $O/p$i
done
