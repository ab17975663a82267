# CPU seconds of a bench run with and without the blocking-sync device flag
cd $GRAFT_REPO_ROOT
for v in 1 0; do
FASIM_BLOCKING_SYNC=$v FASIM_DEBUG_SYNCFLAG=1 python3 - <<'PY'
import os, subprocess, sys, json, resource, time
t0 = time.time()
p = subprocess.run([sys.executable, "bench.py", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True)
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
line = [l for l in p.stdout.splitlines() if l.startswith("{")]
flag = [l for l in p.stderr.splitlines() if "[bench]" in l]
d = json.loads(line[-1]) if line else {}
print(f"FASIM_BLOCKING_SYNC={os.environ['FASIM_BLOCKING_SYNC']}: {d.get('value')} Gcells/s, {d.get('ms_per_step')} ms per step; {ru.ru_utime:.1f} user + {ru.ru_stime:.1f} system CPU-seconds in {time.time() - t0:.1f} s wall; {flag}")
if not line: print(p.stderr[-2000:])
PY
done
