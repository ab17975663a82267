set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
# 1. default bench under the kernel trace (steps 2) + its own JSON
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_kt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r02_bench50mb_profiled.json 2> $O/r02_kt.err
cp $(ls $O/r02_kt/*/*kernel_stats.csv | head -1) $O/r02_bench50mb_kernel_stats.csv; rm -rf $O/r02_kt
echo "step 1 done"
# 2. unprofiled second bench lines
timeout -k 10 500 python3 bench.py --steps 3 --warmup 1 > $O/r02_bench50mb.json 2> $O/r02_b1.err; echo "step 2a done"
timeout -k 10 500 python3 bench.py --steps 2 --warmup 1 --dna genome --no-cpu-baseline > $O/r02_bench50mb_genome.json 2> $O/r02_b2.err; echo "step 2b done"
timeout -k 10 600 python3 bench.py --steps 2 --warmup 1 --dna planted --no-cpu-baseline > $O/r02_bench50mb_planted.json 2> $O/r02_b3.err; echo "step 2c done"
timeout -k 10 500 python3 bench.py --steps 1 --warmup 1 --lncrnas 4 --dna-mb 25 --dna genome --no-cpu-baseline > $O/r02_bench_cfg4_4x3kb_25mb.json 2> $O/r02_b4.err; echo "step 2d done"
# 3. PMC traffic: two separate passes, one batch in flight
export FASIM_WORKERS=1 FASIM_SEG_BATCH=1024
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 bench.py --dna-mb 5 --warmup 0 --no-cpu-baseline > $O/bench_pmc.json 2> $O/pmc_f.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 bench.py --dna-mb 5 --warmup 0 --no-cpu-baseline > $O/bench_pmc_w.json 2> $O/pmc_w.err
python3 tools/pmc_traffic.py $O/pmc_f $O/pmc_w $O/bench_pmc.json > $O/r02_pmc_traffic.json; rm -rf $O/pmc_f $O/pmc_w
unset FASIM_WORKERS FASIM_SEG_BATCH
echo "step 3 done"
# 4. CLI end to end on a 250 Mb single-record FASTA (60-column lines, as genome files come)
python3 - <<'PY'
import sys
sys.path.insert(0, "tools")
import synth, importlib.util
spec = importlib.util.spec_from_file_location("m", "fasim-longtarget_amd/__init__.py"); m = importlib.util.module_from_spec(spec); sys.modules["m"] = m; spec.loader.exec_module(m)
d = m.synth_dna(250_000_000, 777)
with open("/tmp/syn250.fa", "wb") as f:
    f.write(b">syn|chrS|1-250000000\n")
    mv = memoryview(d)
    for i in range(0, len(d), 6000000):
        blk = bytes(mv[i:i + 6000000])
        f.write(b"\n".join(blk[j:j + 60] for j in range(0, len(blk), 60)) + b"\n")
PY
cp tests/golden/H19.fa /tmp/H19.fa; mkdir -p /tmp/out250
( cd /tmp && timeout -k 10 300 $GRAFT_REPO_ROOT/fasim-longtarget_amd/fasim -f1 syn250.fa -f2 H19.fa -O out250/ --stats > /dev/null 2> $GRAFT_REPO_ROOT/$O/r02_cli_250mb.txt ); ls -la /tmp/out250 >> $O/r02_cli_250mb.txt; rm -rf /tmp/out250 /tmp/syn250.fa
echo "step 4 done"
for f in r02_bench50mb r02_bench50mb_genome r02_bench50mb_planted r02_bench_cfg4_4x3kb_25mb r02_bench50mb_profiled; do python3 -c "
import json,sys; d=json.load(open('$O/$f.json')); print('$f', d['value'], d['ms_per_step'], d['mbp_per_s_per_lncrna'], d['per_unit'], {k:d['counts'][k] for k in ('units','hazard_units','rev_exact','exact_replays','align_word_reruns')}, d['isolated_kernels']['ms'])"; done
tail -3 $O/r02_cli_250mb.txt | head -2; grep "end to end" $O/r02_cli_250mb.txt
