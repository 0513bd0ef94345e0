cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_search
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_search -o s5 -- python bench_eval.py --what search --iters 3 --k 1000 > gpurun_out/prof_search/run5.log 2>&1
echo rc=$?
