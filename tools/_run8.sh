mkdir -p gpurun_out/r2h
python tools/chain_bench.py conv_fwd conv_dgrad conv_wgrad wgrad_batch 2>&1 | grep -v amdgpu.ids > gpurun_out/r2h/convs.txt; cat gpurun_out/r2h/convs.txt
