cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 300 ./dpr_scale_amd/selftest ) > gpurun_out/selftest_default.log 2>&1; echo "selftest rc=$?"
( DPRHOT_NO_TR=1 timeout 300 ./dpr_scale_amd/selftest ) > gpurun_out/selftest_notr.log 2>&1; echo "notr rc=$?"
for t in 0 1 2 3 4 5; do ( DPRHOT_TILE=$t timeout 300 ./dpr_scale_amd/selftest ) > gpurun_out/selftest_tile$t.log 2>&1; echo "tile$t rc=$?"; done
( timeout 800 python bench_sweep.py --shapes ${SHAPES:-32x256,128x8192,1024x8192,8192x8192,8192x65536} ) 2>&1 | grep "^{" > gpurun_out/sweep.jsonl; echo "sweep rc=$?"
