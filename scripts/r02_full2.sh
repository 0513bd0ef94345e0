# round-2 evidence of the final code: driver bench command, rocprof kernel trace + PMC traffic of bench.py (cfg2), kernel stats of the
# cfg3-per-rank step, PMC (MFMA busy / LDS conflicts) of the large-shape kernels at 8192 x 8192 x 768, full roofline sweep
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 900 python bench.py ) > $OUT/bench_n1.log 2>&1; echo "bench rc=$?"
tail -n 1 $OUT/bench_n1.log > $OUT/bench_n1.json; cut -c1-600 $OUT/bench_n1.json
ARGS="--steps 500 --warmup 50 --repeats 3 --no-cpu-baseline --no-scale-roofline --no-e2e --no-rank-roofline --driver eager"
rm -rf /tmp/prof_trace /tmp/prof_fetch /tmp/prof_write /tmp/prof_rank
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_trace -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS ) > $OUT/prof_trace.log 2>&1; echo "trace rc=$?"
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_fetch -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS ) > $OUT/prof_fetch.log 2>&1; echo "fetch rc=$?"
( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_write -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS ) > $OUT/prof_write.log 2>&1; echo "write rc=$?"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_rank -o rank -- python $GRAFT_REPO_ROOT/scripts/bench_rankstep.py --shapes 128:8:768:8 --eager --reps 50 ) > $OUT/prof_rank.log 2>&1; echo "prof rank rc=$?"
python scripts/prof_summary.py r02_bench_cfg2 --trace $(find /tmp/prof_trace -name "*.db" | head -1) --fetch $(find /tmp/prof_fetch -name "*.db" | head -1) --write $(find /tmp/prof_write -name "*.db" | head -1) --out $OUT/prof_summary | cut -c1-200 | head -6
python scripts/prof_summary.py r02_cfg3rank --trace $(find /tmp/prof_rank -name "*.db" | head -1) --out $OUT/prof_summary | cut -c1-200 | head -8
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --pmc $set -d /tmp/prof_sw$i -o sw -- python $GRAFT_REPO_ROOT/bench_sweep.py --shapes 8192x8192 ) > $OUT/prof_sw$i.log 2>&1; echo "sw$i rc=$?"
  python scripts/pmc_query.py $(find /tmp/prof_sw$i -name "*.db" | head -1) > $OUT/pmc_sw$i.txt
done
( timeout 900 python bench_sweep.py ) > $OUT/sweep_all.log 2>&1; echo "sweep rc=$?"
grep "^{" $OUT/sweep_all.log > $OUT/sweep.jsonl; python scripts/show_sweep.py $OUT/sweep.jsonl
( timeout 600 python bench_eval.py ) > $OUT/bench_eval.log 2>&1; echo "eval rc=$?"; grep "^{" $OUT/bench_eval.log | cut -c1-700
