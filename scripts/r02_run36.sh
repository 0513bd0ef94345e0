cd $GRAFT_REPO_ROOT
X="--no-e2e --no-cpu-baseline --no-scale-roofline --no-rank-roofline"
for i in 1 2; do timeout 300 python bench.py $X 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['driver'], d['other_driver'], d['timing']['ms_per_step_min'], d['timing']['ms_per_step_max'])"; done
timeout 300 python bench.py $X --driver eager 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('eager', d['value'], d['ms_per_step'])"
DPRHOT_FORCE_DIST=1 timeout 300 python bench.py $X --steps 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('force dist', d['value'], d['ms_per_step'], d.get('torch_distributed_collectives'))"
timeout 300 python -m pytest tests/test_bench_contract.py -q 2>&1 | tail -1
