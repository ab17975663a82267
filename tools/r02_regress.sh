set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; O=gpurun_out
if [ -z "$SKIP_PYTEST" ]; then timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r02_gputest_full.log 2>&1 || true; tail -3 $O/r02_gputest_full.log; fi
# two ranks on the one GPU over gloo: the sharded multi-lncRNA bench path (gather with grouped send/recv, rebase in place)
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --share-gpu --dna-mb 4 --steps 1 --warmup 1 --shard --lncrnas 2 --dna genome --no-cpu-baseline > $O/r02_rehearse_2rank_shard.json 2> $O/r02_rehearse.err || { tail -5 $O/r02_rehearse.err; exit 1; }
python3 -c "
import json; d=json.loads([l for l in open('$O/r02_rehearse_2rank_shard.json') if l.startswith('{')][-1]); print('2-rank rehearsal:', d['n_gpus'], d['value'], d['config']['records_rank0'], d['config']['sharded_record'])"
# one rank, same record (8 Mb), for the record count
timeout -k 10 300 python bench.py --dna-mb 8 --steps 1 --warmup 0 --lncrnas 2 --dna genome --no-cpu-baseline > $O/r02_rehearse_1rank.json 2>> $O/r02_rehearse.err
python3 -c "
import json; d=json.load(open('$O/r02_rehearse_1rank.json')); print('1-rank same record :', d['n_gpus'], d['value'], d['config']['records_rank0'])"
# CLI end to end, 250 Mb, with the parallel tail
python3 - <<'PY'
import sys, importlib.util
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
( cd /tmp && timeout -k 10 300 $GRAFT_REPO_ROOT/fasim-longtarget_amd/fasim -f1 syn250.fa -f2 H19.fa -O out250/ --stats > /dev/null 2> $GRAFT_REPO_ROOT/$O/r02_cli_250mb_v2.txt ); ls -la /tmp/out250 >> $O/r02_cli_250mb_v2.txt; rm -rf /tmp/out250 /tmp/syn250.fa
grep "end to end" $O/r02_cli_250mb_v2.txt
