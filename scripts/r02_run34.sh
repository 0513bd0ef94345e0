cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests -m gpu -q -x -k "search or topk or retriev" ) 2>&1 | tail -2
timeout 300 python bench_eval.py --what search --iters 8 --k 1000 2>&1 | tail -1 | cut -c1-330
timeout 300 python bench_eval.py --what search --iters 8 --k 300 2>&1 | tail -1 | cut -c1-330
timeout 300 python bench_eval.py --what search --iters 8 2>&1 | tail -1 | cut -c1-330
