cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do timeout 300 python bench_eval.py --what search --iters 8 2>&1 | tail -1 | cut -c1-400; done > gpurun_out/search_k100.jsonl
timeout 300 python bench_eval.py --what search --iters 8 --k 1000 2>&1 | tail -1 | cut -c1-400 > gpurun_out/search_k1000.jsonl
cat gpurun_out/search_k100.jsonl gpurun_out/search_k1000.jsonl
bash scripts/r02_final_check.sh
