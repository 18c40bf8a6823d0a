mkdir -p gpurun_out/r2m
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -k "conv2d or wide or bf16 or config3" > gpurun_out/r2m/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r2m/pytest.log
for c in wgrad6430 wgrad12830 wgrad51230; do python tools/kbench.py $c 20; DCV_MFMA_WGRAD9=0 python tools/kbench.py $c 20; done 2>&1 | grep -v amdgpu.ids
