cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export DPRHOT_DIST_BACKEND=gloo DPRHOT_SAME_DEVICE=1
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 50 --warmup 5 ) > gpurun_out/bench_w2.log 2>&1; echo "w2 rc=$?"
tail -5 gpurun_out/bench_w2.log | cut -c1-600
