cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "corpus_search or topk" ) > gpurun_out/pytest_search.log 2>&1; echo "pytest rc=$?"
grep "passed\|failed\|Error" gpurun_out/pytest_search.log | tail -3
( timeout 300 python bench_eval.py --what search ) 2>&1 | grep "^{" | cut -c1-500
( DPRHOT_NO_NL=1 timeout 300 python bench_eval.py --what search ) 2>&1 | grep "^{" | cut -c1-300
