cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -x ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep "passed\|failed\|Error" gpurun_out/pytest_gpu.log | tail -4 | cut -c1-300
python scripts/bench_rankstep.py --shapes 32:8:768:8,32:16:768:8,64:8:768:8,128:8:768:8 --reps 100 2>&1 | grep "^{" | cut -c1-120
