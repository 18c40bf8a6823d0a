mkdir -p gpurun_out/r2q
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2q/bench -o bench -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --no-parity > $R/gpurun_out/r2q/bench_prof.log 2>&1; echo "bench prof rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r2q/pmc_fetch -o f -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-parity > $R/gpurun_out/r2q/pmc_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r2q/pmc_write -o w -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-parity > $R/gpurun_out/r2q/pmc_write.log 2>&1; echo "write rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2q/c100 -o c100 -- python3 $R/tools/wide_bench.py cifar100_deep_model 512 10 > $R/gpurun_out/r2q/c100.log 2>&1; echo "c100 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2q/r18 -o r18 -- python3 $R/tools/wide_bench.py resnet18_style_model 64 10 > $R/gpurun_out/r2q/r18.log 2>&1; echo "r18 rc=$?"
for c in conv6430 conv12830 wgrad6430 wgrad12830; do
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/r2q/mfma_$c -o m -- python3 $R/tools/kbench.py $c 10 > $R/gpurun_out/r2q/mfma_$c.log 2>&1; echo "$c rc=$?"
done
cd $R && python tools/pmc_traffic.py gpurun_out/r2q/pmc_fetch/f_counter_collection.csv gpurun_out/r2q/pmc_write/w_counter_collection.csv gpurun_out/r2q/traffic.json r02 9d91499
find gpurun_out/r2q -name "*stats*.csv" | head; tail -2 gpurun_out/r2q/c100.log gpurun_out/r2q/r18.log
