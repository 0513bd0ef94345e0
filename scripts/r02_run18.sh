cd $GRAFT_REPO_ROOT
for c in 128 64; do echo "DPRHOT_SK_COLS=$c"; DPRHOT_SK_COLS=$c python scripts/bench_rankstep.py --shapes 32:8:768:8,64:2:1024:8,32:8:768:4 --reps 100 2>&1 | grep "^{" | cut -c1-120; done
for ks in 2 3 5; do echo "DPRHOT_SK_KSTEPS=$ks"; DPRHOT_SK_KSTEPS=$ks python scripts/bench_rankstep.py --shapes 32:8:768:8 --reps 100 2>&1 | grep "^{" | cut -c1-120; done
