cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_rank gpurun_out/prof_rank_summary
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_rank -o rank -- python $GRAFT_REPO_ROOT/scripts/bench_rankstep.py --shapes ${SHAPES:-128:8:768:8} --eager --reps 50 ) > gpurun_out/prof_rank.log 2>&1; echo "prof rc=$?"
python scripts/prof_summary.py r02_rank_tmp --trace $(find gpurun_out/prof_rank -name "*.db" | head -1) --out gpurun_out/prof_rank_summary; cut -c1-200 gpurun_out/prof_rank_summary/*kernel_stats*.csv | head -12
