set -o pipefail
mkdir -p gpurun_out/r2e
python tools/chain_bench.py > gpurun_out/r2e/chain.txt 2>&1; echo "chain rc=$?"; cat gpurun_out/r2e/chain.txt
python -m pytest tests/test_ops_gpu.py -m gpu -q -k "conv_norm_layer or fuses or fused" > gpurun_out/r2e/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2e/pytest.log
python bench.py --steps 100 --warmup 10 --no-secondary --no-cpu-baseline > gpurun_out/r2e/bench.json 2> gpurun_out/r2e/bench.err; echo "bench rc=$?"
python -c "
import json; r=json.loads(open('gpurun_out/r2e/bench.json').read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['parity']['max_rel_loss_err'], r['roofline']['avg_launch_us'])"
