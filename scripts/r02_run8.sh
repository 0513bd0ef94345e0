cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -x ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/pytest_gpu.log | grep -v "^$" | tail -25 | cut -c1-250
