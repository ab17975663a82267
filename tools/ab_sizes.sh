# adaptive batch shape against fixed batches of 384 segments at other record sizes (bench.py --dna-mb), Gcells/s and ms per step
cd $GRAFT_REPO_ROOT
for mb in 12 20 33 100; do for ad in 0 1; do
FASIM_ADAPT=$ad timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 --dna-mb $mb --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('dna-mb $mb adapt $ad:', d['value'], 'Gcells/s', d['ms_per_step'], 'ms')"
done; done
