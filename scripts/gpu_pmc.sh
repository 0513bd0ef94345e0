cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
ARGS="${BENCH_ARGS:---steps 200 --warmup 20 --no-cpu-baseline --driver eager}"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --pmc $set -d /tmp/prof_sq$i -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS ) > $OUT/prof_sq$i.log 2>&1; echo "sq$i rc=$?"
  python scripts/pmc_query.py /tmp/prof_sq$i/bench_results.db > $OUT/pmc_sq$i.txt
done
