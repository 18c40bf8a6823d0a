#!/bin/bash
# A/B of one environment switch on the bench step, same box, alternating runs.  usage: gpu_ab.sh VAR
set -o pipefail
mkdir -p gpurun_out/ab
for i in 1 2 3; do
  for v in 1 0; do
    env $1=$v timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --no-parity --steps 400 > gpurun_out/ab/b_${v}_$i.json 2> gpurun_out/ab/b_${v}_$i.err || { echo "bench failed ($1=$v)"; tail -5 gpurun_out/ab/b_${v}_$i.err; exit 1; }
    echo "$1=$v run $i: $(python -c "import json,sys; d=json.loads(open('gpurun_out/ab/b_${v}_$i.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")"
  done
done
