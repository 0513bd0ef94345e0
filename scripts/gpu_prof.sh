# rocprofv3 kernel-trace summary of bench.py (N=1) + PMC passes for HBM traffic; outputs -> gpurun_out/prof_*
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
ARGS="${BENCH_ARGS:---steps 500 --warmup 50 --no-cpu-baseline --no-scale-roofline --driver eager}"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS ) > $OUT/prof_trace.log 2>&1; echo "trace rc=$?"
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_fetch -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS ) > $OUT/prof_fetch.log 2>&1; echo "fetch rc=$?"
( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_write -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS ) > $OUT/prof_write.log 2>&1; echo "write rc=$?"
find $OUT/prof_trace $OUT/prof_fetch $OUT/prof_write -type f | head -30
