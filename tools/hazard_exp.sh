# timing of the phases of the chunked hazard re-run (second batch of a process: no first-launch effects)
cd $GRAFT_REPO_ROOT
for hw in 12 20 30; do for cc in 200 320; do
echo "== hot_w $hw chunk target $cc"
FASIM_HAZARD_HOT_W=$hw FASIM_HAZARD_CHUNK_COLS=$cc FASIM_DEBUG_HAZARD=1 timeout -k 10 120 python tools/hazard_debug.py 2>&1 | awk '/second/{f=1} f'
done; done
