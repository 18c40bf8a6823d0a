#!/bin/bash
# prologue experiments of the persistent launches (tools/variants/resident_prologue.py): phase stamps of block 40 with the kernel arguments in host / device
# memory, with sub-stamps in the backward prologue, and with a forward launch which skips the mailbox read
set -o pipefail
O=gpurun_out/r3p; mkdir -p $O
V=tools/bin/var
run() { name=$1; shift; env "$@" BLOCK=40 timeout -k 10 200 python tools/resident_stamps.py 25 > $O/$name.txt 2>$O/$name.err || { tail -5 $O/$name.err; exit 1; }; echo "== $name"; grep -E "stage input|parameters|total|^--- |^[0-9 ]" $O/$name.txt | head -12; }
run stock &&
run kernarg_host HIP_FORCE_DEV_KERNARG=0 &&
run kernarg_dev HIP_FORCE_DEV_KERNARG=1 &&
run nomail DCV_LIB=$V/libnomail.so &&
{ env DCV_LIB=$V/libbwdsub.so BLOCK=40 timeout -k 10 200 python tools/resident_stamps.py 25 > $O/bwdsub.txt 2>$O/bwdsub.err; echo "== bwdsub"; sed -n '/backward/,$p' $O/bwdsub.txt | head -8; }
