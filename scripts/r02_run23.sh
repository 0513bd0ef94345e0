cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests -m gpu -q -x -k "search or topk or retriev" ) > gpurun_out/pytest_search.log 2>&1; echo "pytest search rc=$?"
tail -2 gpurun_out/pytest_search.log | cut -c1-200
for h in 1024 0 4096; do
  export DPRHOT_SEARCH_HEAD=$h
  echo "head=$h"
  timeout 300 python bench_eval.py --what search --iters 8 2>&1 | tail -1 | cut -c1-400
done
unset DPRHOT_SEARCH_HEAD
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_search
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_search -o s2 -- python bench_eval.py --what search --iters 4 > gpurun_out/prof_search/run2.log 2>&1
echo rc=$?
