cd $GRAFT_REPO_ROOT
echo default; python scripts/bench_rankstep.py --shapes 32:8:768:2,32:8:768:4 --reps 100 2>&1 | grep "^{" | cut -c1-110
echo "skinny from 512 (small step off)"; DPRHOT_NO_SMALL_STEP=1 DPRHOT_SKINNY_MIN_NC=512 python scripts/bench_rankstep.py --shapes 32:8:768:2,32:8:768:4 --reps 100 2>&1 | grep "^{" | cut -c1-110
echo "three-launch path (small step off, no skinny)"; DPRHOT_NO_SMALL_STEP=1 DPRHOT_NO_SKINNY=1 python scripts/bench_rankstep.py --shapes 32:8:768:2,32:8:768:4 --reps 100 2>&1 | grep "^{" | cut -c1-110
