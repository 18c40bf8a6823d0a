for l in tools/bin/var/lib*.so; do echo -n "$(basename $l) "; DCV_LIB=$l python tools/bf16_bench.py 20 256,16,256 512,8,512 2>&1 | tail -2 | cut -c1-75 | tr '\n' ' '; echo; done
