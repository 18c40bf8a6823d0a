set -o pipefail
mkdir -p gpurun_out/r2g
python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r2g/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2g/pytest.log
tail -6 gpurun_out/r2g/pytest.log
python bench.py --steps 100 --warmup 10 --no-secondary > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err; echo "bench rc=$?"
python -c "
import json; r=json.loads(open('gpurun_out/r2g/bench.json').read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['parity']['max_rel_loss_err'], r['roofline']['avg_launch_us'])"
