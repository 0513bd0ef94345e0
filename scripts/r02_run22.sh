cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_search
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_search -o s -- python bench_eval.py --what search --iters 4 > gpurun_out/prof_search/run.log 2>&1
echo rc=$?
find gpurun_out/prof_search -name "*kernel_stats*" | head
f=$(find gpurun_out/prof_search -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-260
