cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests -m gpu -q -x -k "search or topk or retriev" ) 2>&1 | tail -1
for i in 1 2; do timeout 300 python bench_eval.py --what search --iters 8 2>&1 | tail -1 | cut -c1-330; done
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_search
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_search -o s4 -- python bench_eval.py --what search --iters 4 > gpurun_out/prof_search/run4.log 2>&1
