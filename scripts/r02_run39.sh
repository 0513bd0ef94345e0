cd $GRAFT_REPO_ROOT
X="--no-e2e --no-cpu-baseline --no-scale-roofline --no-rank-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["driver"], d["value"], d["ms_per_step"], d["timing"]["ms_per_step_min"], d["timing"]["ms_per_step_max"], d["other_driver"])'
timeout 300 python bench.py $X 2>&1 | tail -1 | python -c "$P"
timeout 300 python bench.py $X --steps 25 --warmup 7 2>&1 | tail -1 | python -c "$P"
timeout 300 python bench.py $X --driver graph10 --steps 20 --warmup 5 2>&1 | tail -1 | python -c "$P"
timeout 300 python bench.py $X --driver eager 2>&1 | tail -1 | python -c "$P"
