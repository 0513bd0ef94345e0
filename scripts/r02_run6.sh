cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/pytest_gpu.log | tail -25
( timeout 600 python bench.py --no-e2e --no-cpu-baseline --no-scale-roofline ) > gpurun_out/bench_quick.log 2>&1; echo "bench rc=$?"
tail -c 2500 gpurun_out/bench_quick.log | grep -v amdgpu.ids
