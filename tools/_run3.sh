set -o pipefail
mkdir -p gpurun_out/r2c
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probes/floor_probe.hip -o gpurun_out/r2c/floor_probe && timeout -k 5 120 gpurun_out/r2c/floor_probe > gpurun_out/r2c/floor_probe.txt 2>&1; echo "probe rc=$?"
cat gpurun_out/r2c/floor_probe.txt
python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/r2c/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2c/pytest.log
tail -8 gpurun_out/r2c/pytest.log
