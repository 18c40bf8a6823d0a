#!/bin/bash
set -o pipefail
O=gpurun_out/r3p; mkdir -p $O
V=tools/bin/var
BLOCK=41 timeout -k 10 200 python tools/resident_stamps.py 25 > $O/stock41.txt 2>$O/stock41.err && sed -n '/backward/,$p' $O/stock41.txt | head -9 &&
DCV_LIB=$V/libnofence.so BLOCK=40 timeout -k 10 200 python tools/resident_stamps.py 25 > $O/nofence40.txt 2>$O/nofence40.err && sed -n '/backward/,$p' $O/nofence40.txt | head -9 && tail -1 $O/nofence40.txt &&
for i in 1 2; do
timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline > $O/bench_stock$i.json 2> $O/bench.err && python -c "import json;d=json.load(open('$O/bench_stock$i.json'));print('stock', d['ms_per_step'], d['roofline'])" &&
DCV_LIB=$V/libnofence.so timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline > $O/bench_nofence$i.json 2> $O/bench.err && python -c "import json;d=json.load(open('$O/bench_nofence$i.json'));print('nofence', d['ms_per_step'], d['roofline'])" || exit 1
done
