for mode in 1 2; do for c in wgrad6430 wgrad12830 wgrad51230; do echo -n "mode $mode: "; DCV_MFMA_WGRAD9_MODE=$mode python tools/kbench.py $c 20 2>&1 | grep -v amdgpu.ids; done; done
for c in wgrad6430 wgrad12830; do echo -n "bf16 "; DCV_KBENCH_MATH=1 python tools/kbench.py $c 20 2>&1 | grep -v amdgpu.ids; done
