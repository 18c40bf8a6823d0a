#!/bin/bash
# round 4: the bench line, the bf16 entry points and the wide models (no tests); logs under gpurun_out/r4/
set -o pipefail
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cut -c1-600 $O/bench.json
python tools/bf16_bench.py 20 ${CASES:-64,32,64 128,16,128 256,16,256 512,8,512} 2>&1 | tail -6 | cut -c1-250 | tee $O/bf16_bench.txt
python tools/wide_bench.py cifar100_deep_model 512 20 graph,bf16 2>&1 | tail -1 | tee $O/wide_c100.txt
python tools/wide_bench.py resnet18_style_model 64 20 graph,bf16 2>&1 | tail -1 | tee $O/wide_r18.txt
