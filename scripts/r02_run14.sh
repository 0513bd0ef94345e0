cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -x ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep "passed\|failed\|Error" gpurun_out/pytest_gpu.log | tail -4 | cut -c1-300
