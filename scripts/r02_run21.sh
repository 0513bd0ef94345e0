cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests -m gpu -q -x -k "search or topk" ) > gpurun_out/pytest_search.log 2>&1; echo "pytest search rc=$?"
tail -2 gpurun_out/pytest_search.log | cut -c1-200
for v in 0 1; do
  if [ $v = 1 ]; then export DPRHOT_SEARCH_LONG_HEAD=1; else unset DPRHOT_SEARCH_LONG_HEAD; fi
  echo "long_head=$v"
  timeout 300 python bench_eval.py --what search --iters 8 2>&1 | tail -2 | cut -c1-600
done
unset DPRHOT_SEARCH_LONG_HEAD
bash scripts/r02_final_check.sh
