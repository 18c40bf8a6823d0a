DCV_KBENCH_NSTAMPS=8 DCV_LIB=tools/bin/var/libwinstamps.so DCV_KBENCH_STAMPS=768 python tools/kbench.py conv6430p 10 192 | grep -v "^CU\|entry\|exit times of"
DCV_KBENCH_NSTAMPS=8 DCV_LIB=tools/bin/var/libwinstamps.so DCV_KBENCH_STAMPS=256 python tools/kbench.py conv6430p 10 64 | grep -v "^CU\|entry\|exit times of"
DCV_KBENCH_NSTAMPS=8 DCV_LIB=tools/bin/var/libwinstamps.so DCV_KBENCH_STAMPS=2048 python tools/kbench.py conv6430p 10 512 | grep -v "entry\|exit times of"
