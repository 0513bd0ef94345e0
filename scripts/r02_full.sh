# round-2 evidence run: full GPU test suite, the driver's bench command, rocprof kernel trace + PMC traffic of bench.py
# and of the cfg3-per-rank step.  Outputs -> gpurun_out/ (copied into profiles/ by hand afterwards).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 1200 python -m pytest tests -m gpu -q -x ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids $OUT/pytest_gpu.log | tail -5 | cut -c1-300
( timeout 900 python bench.py ) > $OUT/bench_n1.log 2>&1; echo "bench rc=$?"
tail -n 1 $OUT/bench_n1.log > $OUT/bench_n1.json; cut -c1-1500 $OUT/bench_n1.json
ARGS="--steps 500 --warmup 50 --repeats 3 --no-cpu-baseline --no-scale-roofline --no-e2e --no-rank-roofline --driver eager"
rm -rf $OUT/prof_trace $OUT/prof_fetch $OUT/prof_write $OUT/prof_rank
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS ) > $OUT/prof_trace.log 2>&1; echo "trace rc=$?"
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_fetch -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS ) > $OUT/prof_fetch.log 2>&1; echo "fetch rc=$?"
( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_write -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS ) > $OUT/prof_write.log 2>&1; echo "write rc=$?"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_rank -o rank -- python $GRAFT_REPO_ROOT/scripts/bench_rankstep.py --shapes 128:8:768:8 --eager --reps 50 ) > $OUT/prof_rank.log 2>&1; echo "prof rank rc=$?"
find $OUT/prof_trace $OUT/prof_fetch $OUT/prof_write $OUT/prof_rank -type f | head -30
python scripts/prof_summary.py r02_bench_cfg2 --trace $(find $OUT/prof_trace -name "*.db" | head -1) --fetch $(find $OUT/prof_fetch -name "*.db" | head -1) --write $(find $OUT/prof_write -name "*.db" | head -1) --out $OUT/prof_summary | cut -c1-220 | head -12
python scripts/prof_summary.py r02_cfg3rank --trace $(find $OUT/prof_rank -name "*.db" | head -1) --out $OUT/prof_summary | cut -c1-220 | head -12
du -sh $OUT/prof_* ; rm -rf $OUT/prof_trace $OUT/prof_fetch $OUT/prof_write $OUT/prof_rank
