cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 ./scratch/sk_timing | tail -4
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "production or packed or whole_step or properties" ) > gpurun_out/pytest_sk.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/pytest_sk.log | tail -5
( timeout 300 python scripts/bench_rankstep.py ) 2>&1 | grep "^{" > gpurun_out/rankstep_new.jsonl; echo "rc=$?"
cat gpurun_out/rankstep_new.jsonl
