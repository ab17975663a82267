# phases of the chunked hazard re-run on one batch of 384 segments alone on the chip (second batch of a process: no
# first-launch effects), the self-check of the windowed checkpoint pass, and the whole-unit run for comparison
cd $GRAFT_REPO_ROOT
echo "== self-check of the windowed checkpoint pass (FASIM_DEBUG_HAZARD=2)"
FASIM_DEBUG_HAZARD=2 timeout -k 10 120 python tools/hazard_debug.py 2>&1 | grep -E "self-check|rror"
echo "== defaults"
FASIM_DEBUG_HAZARD=1 timeout -k 10 120 python tools/hazard_debug.py 2>&1 | awk '/second/{f=1} f'
echo "== checkpoint pass from column 0 (FASIM_HAZARD_SNAP=0)"
FASIM_HAZARD_SNAP=0 FASIM_DEBUG_HAZARD=1 timeout -k 10 120 python tools/hazard_debug.py 2>&1 | awk '/second/{f=1} f'
for hw in 1 8; do for cc in 128 320; do
echo "== hot_w $hw chunk target $cc"
FASIM_HAZARD_HOT_W=$hw FASIM_HAZARD_CHUNK_COLS=$cc FASIM_DEBUG_HAZARD=1 timeout -k 10 120 python tools/hazard_debug.py 2>&1 | awk '/second/{f=1} f'
done; done
