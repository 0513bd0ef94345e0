cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
SH="${SHAPES:-4096x8192}"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --pmc $set -d /tmp/prof_sw$i -o sw -- python $GRAFT_REPO_ROOT/bench_sweep.py --shapes $SH ) > $OUT/prof_sw$i.log 2>&1; echo "sw$i rc=$?"
  python scripts/pmc_query.py /tmp/prof_sw$i/sw_results.db > $OUT/pmc_sw$i.txt
done
