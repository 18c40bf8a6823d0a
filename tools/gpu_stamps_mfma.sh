DCV_KBENCH_NSTAMPS=8 DCV_LIB=tools/bin/var/libwinstamps.so DCV_KBENCH_STAMPS=2048 python tools/kbench.py conv6430p 10 512 | grep -v "entry\|exit times of"
DCV_LIB=tools/bin/var/libgstamps.so DCV_KBENCH_WSTAMPS=512 python tools/kbench.py wgrad6430 10
