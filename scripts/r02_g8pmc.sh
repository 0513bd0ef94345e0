# PMC counters of scratch/g8probe's timing section (8-phase kernel variants), per kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $set -d /tmp/prof_g8_$i -o g8 -- $GRAFT_REPO_ROOT/scratch/g8probe time ) > $OUT/prof_g8_$i.log 2>&1; echo "g8 pmc $i rc=$?"
  python scripts/pmc_query.py $(find /tmp/prof_g8_$i -name "*.db" | head -1) > $OUT/pmc_g8_$i.txt
done
cat $OUT/pmc_g8_1.txt $OUT/pmc_g8_2.txt | cut -c1-200
