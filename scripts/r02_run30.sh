cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -q -x -k "packed or cfg3 or production or eight_ranks or two_ranks" ) 2>&1 | tail -2
for v in 0 1; do
  if [ $v = 1 ]; then export DPRHOT_SK_NO_REMAP=1; else unset DPRHOT_SK_NO_REMAP; fi
  echo "no_remap=$v"
  timeout 200 python scripts/bench_rankstep.py --shapes 128:8:768:8,32:8:768:8 --reps 30 2>&1 | tail -2 | cut -c1-200
done
unset DPRHOT_SK_NO_REMAP
timeout 120 ./scratch/sk_timing | grep "it3 sim" -A1
