#!/bin/bash
# Record lengths around the segmenting edges (cut 5000, step 4900) and tiny records, each against the reference CLI.
cd "$(dirname "$0")/../.."
out=gpurun_out/parity_edges.log
: > $out
for n in 25 60 300 4899 4900 4901 4999 5000 5001 9799 9800 9801 9899 9900 9901 14700 14801; do
  python tests/parity/parity_big.py planted $n $((n + 7)) -lg 20 >> $out 2>&1 || echo "FAILED n=$n" >> $out
done
for n in 300 5000 9901; do
  FASIM_PARITY_RNA=tests/golden/h19_100.fa python tests/parity/parity_big.py planted $n $((n + 3)) -lg 20 >> $out 2>&1 || echo "FAILED n=$n (100-nt query)" >> $out
done
grep -c "identical=True" $out
grep -v "^\[fasim\]" $out | grep -v "identical=True" | head -20
