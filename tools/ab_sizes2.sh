# batch shapes at 20 and 33 Mb: "<seg_batch> <taper>"
cd $GRAFT_REPO_ROOT
run() { FASIM_SEG_BATCH=$2 FASIM_TAPER=$3 timeout -k 10 400 python3 bench.py --steps 4 --warmup 1 --dna-mb $1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('dna-mb $1 seg_batch $2 taper $3:', d['value'], 'Gcells/s', d['ms_per_step'], 'ms')"; }
for cfg in "20 384 0" "20 409 0" "20 409 25" "20 205 25" "20 205 0" "20 272 25" "33 384 0" "33 337 25" "33 337 0" "33 449 25" "33 512 25" "33 225 25"; do run $cfg; done
