#!/bin/bash
# The ONE GPU-side driver script (run through gpurun from the repo root of the snapshot):
#   gpurun --timeout 1500 -- 'bash scripts/gpu_run.sh <task> [<task> ...]'
# Everything lands under gpurun_out/ (merged back); summaries worth keeping are copied into profiles/ by hand.
# Tasks (profiles/README.md lists which file came from which task):
#   tests            pytest -m gpu                                  -> gpurun_out/pytest_gpu.txt
#   tests:<expr>     pytest -m gpu -k <expr>
#   smoke            __graft_entry__.smoke()
#   bench            python bench.py (the driver's command)         -> gpurun_out/bench_n1.json
#   bench:<args>     python bench.py <args> (use + for spaces)
#   prof-bench       rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of the cfg2 bench loop -> gpurun_out/prof_summary/
#   prof-rank        the same three passes of the cfg3-per-rank step (scripts/bench_rankstep.py)   -> gpurun_out/prof_rank_summary/
#   prof-router      the same three passes of the router-width step (bench.py --only router)               -> gpurun_out/prof_router_summary/
#   prof-8192        the same three passes of the 8192^2 x 768 launches (bench_sweep.py)                   -> gpurun_out/prof_big_summary/
#   prof-op          kernel trace of the autograd operator loop (bench.py --only operator)        -> gpurun_out/prof_op_summary/
#   pmc8192          SQ counters (MFMA busy, LDS conflicts, waits) of the 8192^2 x 768 launches    -> gpurun_out/pmc_8192.txt
#   pmc-engine       SQ counters + HBM bytes + durations of scripts/pmc_engine128.py (library GEMM, the 128 x 128 engine's dQ / dC, torch's) -> gpurun_out/pmc_engine128.txt
#   sweep / eval     bench_sweep.py / bench_eval.py                 -> gpurun_out/sweep.jsonl, eval_search.jsonl
#   rank             scripts/bench_rankstep.py (per-rank steps on one GPU) -> gpurun_out/rankstep.jsonl
#   dist1            the multi-rank code paths on one GPU: one-rank RCCL world, two gloo ranks sharing the device
#   fuzz             scripts/fuzz_step.py + scripts/fuzz_topk.py, seed 0
#   cmd:<shell>      any command (use + for spaces)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p "$OUT"

three_passes() {  # <name> <summary tag> <command...>: kernel trace + the two traffic counters, each in its own run
  local name=$1 tag=$2; shift 2
  rm -rf /tmp/prof_${name}_t /tmp/prof_${name}_f /tmp/prof_${name}_w
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_${name}_t -o p -- "$@" ) > $OUT/prof_${name}_trace.log 2>&1; echo "$name trace rc=$?"
  ( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_${name}_f -o p -- "$@" ) > $OUT/prof_${name}_fetch.log 2>&1; echo "$name fetch rc=$?"
  ( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_${name}_w -o p -- "$@" ) > $OUT/prof_${name}_write.log 2>&1; echo "$name write rc=$?"
  python scripts/prof_summary.py $tag --trace "$(find /tmp/prof_${name}_t -name '*.db' | head -1)" --fetch "$(find /tmp/prof_${name}_f -name '*.db' | head -1)" \
      --write "$(find /tmp/prof_${name}_w -name '*.db' | head -1)" --out $OUT/prof_${name}_summary | cut -c1-220 | head -12
}

for task in "$@"; do
  arg=""; case "$task" in *:*) arg="${task#*:}"; arg="${arg//+/ }"; task="${task%%:*}";; esac
  echo "=== $task $arg"
  case "$task" in
    tests)  ( timeout 1500 python -m pytest tests -m gpu -x -q --timeout 420 ${arg:+-k "$arg"} ) > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -n 15 $OUT/pytest_gpu.txt ;;
    smoke)  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 3 ;;
    bench)  ( timeout 1500 python bench.py $arg ) > $OUT/bench_n1.log 2>&1; echo "bench rc=$?"; tail -n 1 $OUT/bench_n1.log > $OUT/bench_n1.json; cut -c1-1500 $OUT/bench_n1.json ;;
    prof-bench) three_passes bench r06_bench_cfg2 python $GRAFT_REPO_ROOT/bench.py --steps 500 --warmup 50 --repeats 3 --only step --driver eager ;;
    prof-rank)  three_passes rank r06_cfg3rank python $GRAFT_REPO_ROOT/scripts/bench_rankstep.py --shapes ${arg:-128:8:768:8} --eager --reps 50 ;;
    prof-router) three_passes router r06_router python $GRAFT_REPO_ROOT/bench.py --only router ;;
    prof-8192)  three_passes big r06_8192 python $GRAFT_REPO_ROOT/bench_sweep.py --shapes 8192x8192 ;;
    prof-op)    three_passes op r06_operator python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --repeats 3 --only operator ;;
    pmc8192)
      i=0
      for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
        i=$((i+1)); rm -rf /tmp/prof_pmc$i
        ( cd /tmp && timeout 600 rocprofv3 --pmc $set -d /tmp/prof_pmc$i -o p -- python $GRAFT_REPO_ROOT/bench_sweep.py --shapes 8192x8192 ) > $OUT/prof_pmc$i.log 2>&1; echo "pmc$i rc=$?"
        python scripts/pmc_query.py "$(find /tmp/prof_pmc$i -name '*.db' | head -1)" >> $OUT/pmc_8192.txt
      done; cut -c1-200 $OUT/pmc_8192.txt | head -60 ;;
    pmc-engine)
      rm -f $OUT/pmc_engine128.txt; i=0
      for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
        i=$((i+1)); rm -rf /tmp/prof_pe$i
        ( cd /tmp && timeout 600 rocprofv3 --pmc $set -d /tmp/prof_pe$i -o p -- python $GRAFT_REPO_ROOT/scripts/pmc_engine128.py ) > $OUT/prof_pe$i.log 2>&1; echo "pmc-engine$i rc=$?"
        python scripts/pmc_query.py "$(find /tmp/prof_pe$i -name '*.db' | head -1)" >> $OUT/pmc_engine128.txt
      done
      rm -rf /tmp/prof_pet
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_pet -o p -- python $GRAFT_REPO_ROOT/scripts/pmc_engine128.py ) > $OUT/prof_pet.log 2>&1; echo "pmc-engine trace rc=$?"
      python scripts/prof_summary.py r06_engine128 --trace "$(find /tmp/prof_pet -name '*.db' | head -1)" --out $OUT/prof_engine_summary 2>&1 | cut -c1-200 | head -30
      grep -v "elementwise\|copyBuffer\|fillBuffer" $OUT/pmc_engine128.txt | cut -c1-160 | head -150 ;;
    sweep)  ( timeout 900 python bench_sweep.py $arg ) > $OUT/sweep.jsonl 2> $OUT/sweep.err; echo "sweep rc=$?"; python scripts/show_sweep.py $OUT/sweep.jsonl 2>/dev/null | head -30 ;;
    eval)   ( timeout 900 python bench_eval.py $arg ) > $OUT/eval_search.jsonl 2> $OUT/eval.err; echo "eval rc=$?"; cut -c1-400 $OUT/eval_search.jsonl ;;
    rank)   ( timeout 600 python scripts/bench_rankstep.py $arg ) > $OUT/rankstep.jsonl 2> $OUT/rankstep.err; echo "rank rc=$?"; cat $OUT/rankstep.jsonl ;;
    dist1)
      Q="--only step --steps 100 --repeats 5"
      ( DPRHOT_FORCE_DIST=1 timeout 300 python bench.py $Q ) > $OUT/bench_force_dist.log 2>&1; echo "one-rank RCCL world rc=$?"; tail -n 1 $OUT/bench_force_dist.log | cut -c1-700
      ( DPRHOT_DIST_BACKEND=gloo DPRHOT_SAME_DEVICE=1 timeout 600 python bench.py --gpus 2 $Q ) > $OUT/bench_w2_gloo.log 2>&1; echo "2 ranks / 1 device / gloo rc=$?"; tail -n 1 $OUT/bench_w2_gloo.log | cut -c1-700 ;;
    fuzz)   ( timeout 900 python scripts/fuzz_step.py --seed 0 ) > $OUT/fuzz_step.txt 2>&1; echo "fuzz_step rc=$?"; tail -n 3 $OUT/fuzz_step.txt
            ( timeout 900 python scripts/fuzz_topk.py --seed 0 ) > $OUT/fuzz_topk.txt 2>&1; echo "fuzz_topk rc=$?"; tail -n 3 $OUT/fuzz_topk.txt ;;
    cmd)    ( eval "timeout 1200 $arg" ) > $OUT/cmd.log 2>&1; echo "cmd rc=$?"; tail -n 40 $OUT/cmd.log ;;
    *) echo "unknown task $task" ;;
  esac
done
