mkdir -p gpurun_out/r2j
python tools/find_aten_ops.py > gpurun_out/r2j/aten_ops.txt 2>&1; tail -40 gpurun_out/r2j/aten_ops.txt
