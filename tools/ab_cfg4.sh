# BASELINE config 4 in small (4 x 3 kb lncRNAs x 25 Mb as one batch): fixed batches of 384 against the fitted batch shape
cd $GRAFT_REPO_ROOT
for pass in 1 2; do for v in 0 1; do
FASIM_ADAPT_MULTI=$v timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --lncrnas 4 --dna-mb 25 --dna genome --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('pass $pass FASIM_ADAPT_MULTI=$v:', d['value'], 'Gcells/s', d['ms_per_step'], 'ms')"
done; done
