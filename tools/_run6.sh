mkdir -p gpurun_out/r2f
for skip in 0 6 5 3; do echo "== DCV_NC_SKIP=$skip (6: sums only, 5: finalize only, 3: apply only)"; DCV_NC_SKIP=$skip python tools/chain_bench.py norm_fwd norm_bwd 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r2f/stages.txt
cat gpurun_out/r2f/stages.txt
