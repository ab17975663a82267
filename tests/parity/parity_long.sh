#!/bin/bash
# Long-query parity runs (query tiling: MALAT1 = 3 tiles, NEAT1 = 8 tiles, MEG3 = 1 tile) against the reference CLI.
cd "$(dirname "$0")/../.."
out=gpurun_out/parity_long.log
: > $out
FASIM_PARITY_RNA=tests/golden/MALAT1.fa python tests/parity/parity_big.py planted 1500000 71 >> $out 2>&1 &
FASIM_PARITY_RNA=tests/golden/MALAT1.fa python tests/parity/parity_big.py random 1500000 72 -lg 30 >> $out 2>&1 &
FASIM_PARITY_RNA=tests/golden/NEAT1.fa python tests/parity/parity_big.py planted 600000 73 >> $out 2>&1 &
FASIM_PARITY_RNA=tests/golden/NEAT1.fa python tests/parity/parity_big.py random 600000 74 -lg 30 -t 1 >> $out 2>&1 &
FASIM_PARITY_RNA=tests/golden/MEG3.fa python tests/parity/parity_big.py planted 3000000 75 >> $out 2>&1 &
wait
grep -c "identical=True" $out
grep "identical=" $out
