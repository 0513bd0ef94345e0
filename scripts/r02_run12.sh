cd $GRAFT_REPO_ROOT
for ks in 0 4 5 7 9 11; do
  echo "DPRHOT_SK_KSTEPS=$ks"; DPRHOT_SK_KSTEPS=$ks python scripts/bench_rankstep.py --shapes 128:8:768:8 --eager --reps 300 2>&1 | grep "^{" | cut -c1-200
done
for gp in 1 2 4; do
  echo "DPRHOT_SK_GPARTS=$gp"; DPRHOT_SK_GPARTS=$gp python scripts/bench_rankstep.py --shapes 128:8:768:8 --eager --reps 300 2>&1 | grep "^{" | cut -c1-200
done
