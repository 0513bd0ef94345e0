cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "no_logits or score_free or rank_and_loss or properties_full or selftest" ) > gpurun_out/pytest_nl.log 2>&1; echo "pytest rc=$?"
grep "passed\|failed" gpurun_out/pytest_nl.log | tail -3
( timeout 300 python bench_sweep.py --shapes 8192x8192 ) 2>&1 | grep "^{" | python scripts/show_sweep.py
