# is the exclusive k_scan time of the isolated pass stable from run to run? (planted DNA three times, then random, then planted)
cd $GRAFT_REPO_ROOT
for d in planted planted planted random planted; do
timeout -k 10 300 python3 bench.py --steps 1 --warmup 1 --dna $d --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$d', d['ms_per_step'], d['isolated_kernels']['ms']['k_scan (fused stage 1+2)'], d['roofline']['launches'])"
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor (edge|junction)" | head -4
done
