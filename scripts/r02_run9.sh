cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -x ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/pytest_gpu.log | grep -v "^$" | grep "passed\|failed\|Error\|error" | tail -8 | cut -c1-250
( timeout 600 python bench_sweep.py --shapes 1024x8192,8192x8192,8192x65536 ) > gpurun_out/sweep_big.log 2>&1; echo "sweep rc=$?"
grep "^{" gpurun_out/sweep_big.log | python scripts/show_sweep.py 2>/dev/null || grep "^{" gpurun_out/sweep_big.log | cut -c1-900
