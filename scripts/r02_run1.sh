# round 2, call 1: new production-entry parity tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/pytest_gpu.log | tail -30
