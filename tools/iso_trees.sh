# exclusive k_scan time of bench.py's isolated pass (planted DNA) for several source trees on one box
cd $GRAFT_REPO_ROOT
for t in "$@"; do
( cd $t && timeout -k 10 300 python3 bench.py --steps 1 --warmup 1 --dna planted --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$t', d['ms_per_step'], d['isolated_kernels']['ms']['k_scan (fused stage 1+2)'], d['roofline']['launches'])" )
done
