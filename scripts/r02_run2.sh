cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "production or packed or whole_step or properties" ) > gpurun_out/pytest_sk.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/pytest_sk.log | tail -30
( timeout 300 python scripts/bench_rankstep.py ) 2>&1 | grep "^{" > gpurun_out/rankstep_new.jsonl; echo "rc=$?"
( DPRHOT_NO_SKINNY=1 timeout 300 python scripts/bench_rankstep.py ) 2>&1 | grep "^{" > gpurun_out/rankstep_old.jsonl; echo "rc=$?"
cat gpurun_out/rankstep_new.jsonl gpurun_out/rankstep_old.jsonl
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_rank -o rank -- python $GRAFT_REPO_ROOT/scripts/bench_rankstep.py --shapes 128:8:768:8 --eager --reps 50 ) > gpurun_out/prof_rank.log 2>&1; echo "prof rc=$?"
find gpurun_out/prof_rank -type f | head
python scripts/prof_summary.py r02_rank_tmp --trace $(find gpurun_out/prof_rank -name "*.db" | head -1) --out gpurun_out/prof_rank_summary; cat gpurun_out/prof_rank_summary/*kernel_stats*.csv | head -20
