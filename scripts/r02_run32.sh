cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
X="--steps 100 --warmup 20 --repeats 5 --no-e2e --no-cpu-baseline --no-scale-roofline --no-rank-roofline"
echo "== force dist, one RCCL rank"
DPRHOT_FORCE_DIST=1 timeout 300 python bench.py $X 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['n_gpus'], d['config']['collectives'], d.get('torch_distributed_collectives'), d['roofline'] is not None)"
echo "== force dist, watchdog fires"
DPRHOT_FORCE_DIST=1 DPRHOT_DIRECT_RCCL_TIMEOUT=0.0001 timeout 300 python bench.py $X 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['n_gpus'], d['config']['collectives'], d['roofline'])"
echo "== 2 ranks gloo on one device"
DPRHOT_DIST_BACKEND=gloo DPRHOT_SAME_DEVICE=1 timeout 300 python bench.py --gpus 2 $X 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['n_gpus'], d['config']['collectives'], d['rccl_ranks'])"
echo "== gpus 2 on a one-GPU box"
timeout 60 python bench.py --gpus 2 $X; echo "rc=$?"
echo "== default"
timeout 600 python bench.py --no-e2e 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['driver'], d['roofline']['frac'], d['timing'])"
