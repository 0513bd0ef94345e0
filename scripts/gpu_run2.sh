set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 300 ./dpr_scale_amd/selftest time ) > gpurun_out/selftest_default.log 2>&1; echo "selftest rc=$?"
( DPRHOT_NO_TR=1 timeout 300 ./dpr_scale_amd/selftest ) > gpurun_out/selftest_notr.log 2>&1; echo "notr rc=$?"
( DPRHOT_UNFUSED_BWD=1 timeout 300 ./dpr_scale_amd/selftest time ) > gpurun_out/selftest_unfused.log 2>&1; echo "unfused rc=$?"
( DPRHOT_BIG_MIN=1 timeout 300 ./dpr_scale_amd/selftest ) > gpurun_out/selftest_big1.log 2>&1; echo "big1 rc=$?"
( DPRHOT_NO_SMALL_STEP=1 timeout 300 ./dpr_scale_amd/selftest ) > gpurun_out/selftest_nosmall.log 2>&1; echo "nosmall rc=$?"
for t in 0 1 2 3 4 5; do ( DPRHOT_TILE=$t timeout 300 ./dpr_scale_amd/selftest ) > gpurun_out/selftest_tile$t.log 2>&1; echo "tile$t rc=$?"; done
( timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
( timeout 600 python bench.py ) > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
grep -c FAIL gpurun_out/selftest_*.log
