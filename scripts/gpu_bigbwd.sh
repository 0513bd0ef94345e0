cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 300 ./dpr_scale_amd/selftest ) > gpurun_out/selftest_default.log 2>&1; echo "selftest rc=$?"
( DPRHOT_BIG_MIN=1 timeout 300 ./dpr_scale_amd/selftest ) > gpurun_out/selftest_big1.log 2>&1; echo "big1 rc=$?"
grep -c FAIL gpurun_out/selftest_default.log gpurun_out/selftest_big1.log
grep -E "^case|FAIL" gpurun_out/selftest_big1.log | grep -B1 FAIL | head -20
timeout 600 python bench_sweep.py --shapes 1024x8192,8192x8192,8192x65536 2>&1 | grep "^{" > gpurun_out/sweep_big.jsonl; python scripts/show_sweep.py gpurun_out/sweep_big.jsonl
DPRHOT_NO_BIG_BWD=1 timeout 600 python bench_sweep.py --shapes 1024x8192 2>&1 | grep "^{" > gpurun_out/sweep_nobig.jsonl; python scripts/show_sweep.py gpurun_out/sweep_nobig.jsonl
