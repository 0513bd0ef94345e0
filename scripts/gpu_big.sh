cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 300 ./dpr_scale_amd/selftest time ) > gpurun_out/selftest_default.log 2>&1; echo "selftest rc=$?"
( DPRHOT_BIG_MIN=1 timeout 300 ./dpr_scale_amd/selftest ) > gpurun_out/selftest_big1.log 2>&1; echo "big1 rc=$?"
( DPRHOT_BIG_MIN=0 timeout 300 ./dpr_scale_amd/selftest time ) > gpurun_out/selftest_big0.log 2>&1; echo "big0 rc=$?"
grep -c FAIL gpurun_out/selftest_default.log gpurun_out/selftest_big1.log gpurun_out/selftest_big0.log
grep -A3 "^search nq=1024" gpurun_out/selftest_default.log gpurun_out/selftest_big0.log
( DPRHOT_BIG_MIN=1 timeout 600 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -3
( timeout 600 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -3
timeout 600 python bench_sweep.py --shapes 1024x8192,8192x8192,8192x65536 2>&1 | grep "^{" > gpurun_out/sweep_big.jsonl; python scripts/show_sweep.py gpurun_out/sweep_big.jsonl
timeout 400 python bench_eval.py --what both 2>/dev/null
