DCV_LIB=tools/bin/var/libstamps.so DCV_KBENCH_STAMPS=2048 python tools/kbench.py conv6430p 10
DCV_LIB=tools/bin/var/libstamps.so DCV_KBENCH_STAMPS=1024 python tools/kbench.py conv12830p 10
