cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -q -x -k "packed or cfg3 or production or eight_ranks or tiles_over" ) 2>&1 | tail -2
for v in 0 1; do
  if [ $v = 1 ]; then export DPRHOT_SK_NO_DEEP=1; else unset DPRHOT_SK_NO_DEEP; fi
  echo "no_deep=$v"
  timeout 200 python scripts/bench_rankstep.py --shapes 128:8:768:8 --reps 30 2>&1 | tail -1 | cut -c1-200
done
unset DPRHOT_SK_NO_DEEP
timeout 120 ./scratch/sk_timing | grep "it3 sim" -A1
