# rocprofv3 kernel trace of the retrieval flow (bench_eval.py --what search) at k = 100 and k = 1000 -> profiles/r02_search_*_kernel_stats.csv
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/prof_s100 $OUT/prof_s1000
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_s100 -o s -- python $GRAFT_REPO_ROOT/bench_eval.py --what search --iters 4 ) > $OUT/prof_s100.log 2>&1; echo "rc=$?"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_s1000 -o s -- python $GRAFT_REPO_ROOT/bench_eval.py --what search --iters 4 --k 1000 ) > $OUT/prof_s1000.log 2>&1; echo "rc=$?"
python scripts/prof_summary.py r02_search_k100 --trace $(find $OUT/prof_s100 -name "*.db" | head -1) --out $OUT/prof_summary | grep dprhot | cut -c1-200
python scripts/prof_summary.py r02_search_k1000 --trace $(find $OUT/prof_s1000 -name "*.db" | head -1) --out $OUT/prof_summary | grep dprhot | cut -c1-200
grep '"what"' $OUT/prof_s100.log $OUT/prof_s1000.log | cut -c1-260
rm -rf $OUT/prof_s100 $OUT/prof_s1000
