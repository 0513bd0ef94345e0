cd $GRAFT_REPO_ROOT
for ks in 0 10 8 6; do
  if [ $ks = 0 ]; then unset DPRHOT_SK_KSTEPS; else export DPRHOT_SK_KSTEPS=$ks; fi
  echo "ksteps=$ks"
  timeout 200 python scripts/bench_rankstep.py --shapes 128:8:768:8 --reps 30 2>&1 | tail -2 | cut -c1-300
done
