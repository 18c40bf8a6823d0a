#!/bin/bash
# phase-skip variants (tools/bin/var/lib${PAT}*.so) of the bf16 window kernel on the bench shapes
mkdir -p gpurun_out/r4
for l in tools/bin/var/lib${PAT:-k_}*.so; do echo "$(basename $l):"; DCV_LIB=$l python tools/bf16_bench.py 20 ${CASES:-256,16,256 512,8,512 128,16,128} 2>&1 | tail -3 | cut -c1-150; done
