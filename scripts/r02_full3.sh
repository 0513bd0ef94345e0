# final evidence of round 2: driver bench command (r02_v3), multi-rank code paths on one GPU, kernel stats / traffic of bench.py
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 900 python bench.py ) > $OUT/bench_n1.log 2>&1; echo "bench rc=$?"
tail -n 1 $OUT/bench_n1.log > $OUT/bench_n1.json; cut -c1-400 $OUT/bench_n1.json
Q="--no-e2e --no-cpu-baseline --no-scale-roofline --no-rank-roofline --steps 100 --repeats 5"
( timeout 300 python bench.py --gpus 2 $Q ) > $OUT/bench_g2_refuse.log 2>&1; echo "gpus 2 on a 1-GPU box rc=$? (2 = refused)"; tail -n 2 $OUT/bench_g2_refuse.log | cut -c1-300
( DPRHOT_FORCE_DIST=1 timeout 300 python bench.py $Q ) > $OUT/bench_force_dist.log 2>&1; echo "one-rank RCCL world rc=$?"; tail -n 1 $OUT/bench_force_dist.log | cut -c1-700
( DPRHOT_DIST_BACKEND=gloo DPRHOT_SAME_DEVICE=1 timeout 600 python bench.py --gpus 2 $Q ) > $OUT/bench_w2_gloo.log 2>&1; echo "2 ranks / 1 device / gloo rc=$?"; tail -n 1 $OUT/bench_w2_gloo.log | cut -c1-700
ARGS="--steps 500 --warmup 50 --repeats 3 --no-cpu-baseline --no-scale-roofline --no-e2e --no-rank-roofline --driver eager"
rm -rf /tmp/prof_trace /tmp/prof_fetch /tmp/prof_write
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_trace -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS ) > $OUT/prof_trace.log 2>&1; echo "trace rc=$?"
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_fetch -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS ) > $OUT/prof_fetch.log 2>&1; echo "fetch rc=$?"
( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_write -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS ) > $OUT/prof_write.log 2>&1; echo "write rc=$?"
python scripts/prof_summary.py r02_bench_cfg2 --trace $(find /tmp/prof_trace -name "*.db" | head -1) --fetch $(find /tmp/prof_fetch -name "*.db" | head -1) --write $(find /tmp/prof_write -name "*.db" | head -1) --out $OUT/prof_summary | cut -c1-200 | head -4
