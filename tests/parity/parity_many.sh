#!/bin/bash
# Several large parity runs side by side on the GPU box (the reference CLI is single-threaded; the box has 16 cores).
# Each line of output is one comparison of the -TFOsorted bytes: our `fasim` CLI (HIP path) vs oracle/_ref/fasim_ref.
cd "$(dirname "$0")/../.."
out=gpurun_out/parity_many.log
: > $out
run() { python tests/parity/parity_big.py "$@" >> $out 2>&1 & }
run random 3000000 1001
run planted 3000000 1002
run random 2000000 1003 -c 3000 -o 60 -lg 30
run planted 2000000 1004 -r 2 -t 1 -lg 25 -ds 10
run random 2000000 1005 -i 50 -S 0.8 -ni 15 -na 500 -lg 30
run planted 1500000 1006 -pc 2 -pt 3 -t -1 -lg 35
wait
grep -c "identical=True" $out
grep "identical=" $out
