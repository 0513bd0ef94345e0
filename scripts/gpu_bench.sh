cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 300 ./dpr_scale_amd/selftest ) > gpurun_out/selftest_default.log 2>&1; echo "selftest rc=$?"
( timeout 600 python bench.py ${BENCH_ARGS:-} ) > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/bench.log
