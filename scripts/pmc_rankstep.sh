#!/bin/bash
# SQ counters (waits, LDS activity and bank conflicts) of the cfg3-per-rank step's kernels: gpurun -- bash scripts/pmc_rankstep.sh
# NOTE (round 4): three further passes with TA_* / TCP_* / TCC_* counters (texture-path busy and stall cycles, L2 hits and misses)
# did not finish within 300 s each on this pool -- 15 GPU-minutes for nothing; they are not run here.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/pr_sq
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pr_sq -o p -- \
    python $GRAFT_REPO_ROOT/scripts/bench_rankstep.py --shapes 128:8:768:8 --eager --reps 20 ) > /tmp/pr_sq.log 2>&1; echo "pmc rc=$?"
python scripts/pmc_query.py "$(find /tmp/pr_sq -name '*.db' | head -1)" | grep "sk_"
