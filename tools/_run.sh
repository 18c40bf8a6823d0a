timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -12 gpurun_out/gpu_tests.log
for v in 1 2; do python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['config']['final_loss'])"; done
