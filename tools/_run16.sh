set -o pipefail
mkdir -p gpurun_out/r2p
python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r2p/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2p/pytest.log
tail -6 gpurun_out/r2p/pytest.log
python bench.py --steps 100 --warmup 10 > gpurun_out/r2p/bench.json 2> gpurun_out/r2p/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r2p/bench.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['parity']['max_rel_loss_err'], r['roofline']['avg_launch_us'], r['cpu_baseline']['value'])
for s in r['secondary']:
    print(s.get('op',s.get('workload'))[:70], s['math'][:6], s.get('avg_launch_us'), s.get('achieved',s.get('value')), s.get('frac'))
PY
