cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/bench_n1.log 2>&1; echo "bench rc=$?"
tail -c 6000 gpurun_out/bench_n1.log | grep -v amdgpu.ids
( timeout 120 python bench.py --gpus 2 --steps 5 --warmup 2 ) > gpurun_out/bench_n2_refuse.log 2>&1; echo "bench --gpus 2 on one GPU rc=$? (expect 2)"; tail -3 gpurun_out/bench_n2_refuse.log
( DPRHOT_DIST_BACKEND=gloo DPRHOT_SAME_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --repeats 5 ) > gpurun_out/bench_n2_gloo.log 2>&1; echo "bench gloo2 rc=$?"; tail -c 3000 gpurun_out/bench_n2_gloo.log | grep -v amdgpu.ids
