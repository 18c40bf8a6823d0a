#!/bin/bash
# usage: gpu_variants.sh "<kbench case> [<case> ...]" [batch]  — times every library under tools/bin/var/ on the given kbench cases
for c in $1; do for l in tools/bin/var/lib*.so; do echo -n "$(basename $l) "; DCV_LIB=$l python tools/kbench.py $c 30 $2 2>&1 | tail -1; done; done
