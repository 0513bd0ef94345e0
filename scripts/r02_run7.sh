cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_citadel_router.py tests/test_gpu_parity.py -m gpu -q -x ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/pytest_gpu.log | tail -8 | cut -c1-300
( timeout 600 python bench.py --no-e2e --no-cpu-baseline --no-scale-roofline --steps 50 --repeats 5 ) > gpurun_out/bench_quick.log 2>&1; echo "bench rc=$?"
tail -c 800 gpurun_out/bench_quick.log | grep -v amdgpu.ids
