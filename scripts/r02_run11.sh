cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "cfg3 or cfg5 or packed or selftest or properties_full or whole_step" ) > gpurun_out/pytest_sk.log 2>&1; echo "pytest rc=$?"
grep "passed\|failed" gpurun_out/pytest_sk.log | tail -3
python scripts/bench_rankstep.py --shapes 128:8:768:8 --eager --reps 200 2>&1 | tail -3
DPRHOT_SK_COLS=128 python scripts/bench_rankstep.py --shapes 128:8:768:8 --eager --reps 200 2>&1 | tail -2
rm -rf /tmp/prof_rank
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_rank -o rank -- python $GRAFT_REPO_ROOT/scripts/bench_rankstep.py --shapes 128:8:768:8 --eager --reps 50 ) > gpurun_out/prof_rank.log 2>&1; echo "prof rank rc=$?"
python scripts/prof_summary.py r02_cfg3rank --trace $(find /tmp/prof_rank -name "*.db" | head -1) --out gpurun_out/prof_summary | cut -c1-160 | head -6
