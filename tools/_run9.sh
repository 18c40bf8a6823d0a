mkdir -p gpurun_out/r2i
for tb in 256 512 1024 2048 4096; do echo "== DCV_CONV_TARGET_BLOCKS=$tb"; DCV_CONV_TARGET_BLOCKS=$tb python tools/chain_bench.py conv_fwd conv_dgrad 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r2i/target_blocks.txt
cat gpurun_out/r2i/target_blocks.txt
