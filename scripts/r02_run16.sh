cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python scripts/bench_rankstep.py --shapes 128:8:768:8 --eager --reps 300 2>&1 | grep "^{" | cut -c1-200
rm -rf /tmp/prof_rank
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_rank -o rank -- python $GRAFT_REPO_ROOT/scripts/bench_rankstep.py --shapes 128:8:768:8 --eager --reps 50 ) > gpurun_out/prof_rank.log 2>&1; echo "prof rank rc=$?"
python scripts/prof_summary.py r02_cfg3rank --trace $(find /tmp/prof_rank -name "*.db" | head -1) --out gpurun_out/prof_summary | cut -c1-130 | head -6
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "cfg3 or cfg5 or packed or selftest" ) 2>&1 | grep "passed\|failed" | tail -2
