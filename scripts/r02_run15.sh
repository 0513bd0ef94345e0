cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -x ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep "passed\|failed\|Error" gpurun_out/pytest_gpu.log | tail -4 | cut -c1-300
python scripts/bench_rankstep.py --shapes 128:8:768:8 --eager --reps 300 2>&1 | grep "^{" | cut -c1-200
rm -rf /tmp/prof_rank
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_rank -o rank -- python $GRAFT_REPO_ROOT/scripts/bench_rankstep.py --shapes 128:8:768:8 --eager --reps 50 ) > gpurun_out/prof_rank.log 2>&1; echo "prof rank rc=$?"
python scripts/prof_summary.py r02_cfg3rank --trace $(find /tmp/prof_rank -name "*.db" | head -1) --out gpurun_out/prof_summary | cut -c1-130 | head -6
python bench.py --no-e2e --no-cpu-baseline --no-scale-roofline --no-rank-roofline 2>&1 | tail -n 1 | cut -c1-300
python bench_sweep.py --shapes 1024x8192,128x8192 2>&1 | grep "^{" | python scripts/show_sweep.py
