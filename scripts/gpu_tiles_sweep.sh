cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for t in 0 1 2 3 4 5; do echo "== tile $t"; DPRHOT_TILE=$t timeout 300 python bench_sweep.py --shapes ${SHAPES:-128x8192,256x8192} 2>&1 | grep "^{" > gpurun_out/tsweep_$t.jsonl; python scripts/show_sweep.py gpurun_out/tsweep_$t.jsonl; done
