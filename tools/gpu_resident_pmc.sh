#!/bin/bash
# PMC record of the bench workload's two persistent launches: matrix-pipe busy cycles, VALU / LDS activity next to the traffic counters
# (tools/measure_round.sh already records FETCH_SIZE / WRITE_SIZE).  usage: gpu_resident_pmc.sh <commit>
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/respmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-parity"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- $CMD > $O/f.log 2>&1 || { tail -5 $O/f.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- $CMD > $O/w.log 2>&1 || { tail -5 $O/w.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/m -o m -- $CMD > $O/m.log 2>&1 || { tail -5 $O/m.log; exit 1; }
python3 $R/tools/pmc_kernels.py $O/r03_pmc_resident_mfma.json r03 ${1:-?} "rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES> -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-parity (separate passes)" \
  $O/f $O/w $O/m resident_fwd_kernel resident_bwd_kernel > $O/summary.txt || { tail -5 $O/summary.txt; exit 1; }
python3 -c "import json; r=json.load(open('$O/r03_pmc_resident_mfma.json')); [print(k, v) for k, v in r['kernels'].items()]"
rm -rf $O/f $O/w $O/m
