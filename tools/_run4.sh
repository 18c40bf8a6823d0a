set -o pipefail
mkdir -p gpurun_out/r2d
python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2d/pytest.log
tail -12 gpurun_out/r2d/pytest.log
python bench.py --steps 100 --warmup 10 --no-secondary > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err; echo "bench rc=$?"
python -c "
import json; r=json.loads(open('gpurun_out/r2d/bench.json').read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['parity']['max_rel_loss_err'], r['roofline']['avg_launch_us'])"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2d/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --no-parity > $GRAFT_REPO_ROOT/gpurun_out/r2d/prof.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT && find gpurun_out/r2d/prof -name "*kernel_stats.csv" | head
