cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( timeout 1500 python -m pytest tests -m gpu -q -x ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep "passed\|failed\|Error" gpurun_out/pytest_gpu.log | tail -3 | cut -c1-300
( timeout 900 python bench.py ) > gpurun_out/bench_n1.log 2>&1; echo "bench rc=$?"
tail -n 1 gpurun_out/bench_n1.log > gpurun_out/bench_n1.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_n1.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['other_driver'])
at=d['roofline_at_scale']; print({k:(v['us'],v['frac']) for k,v in at.items() if isinstance(v,dict)}, at.get('step_us'))
print(d['roofline_cfg3_rank']['step_us'], d['roofline_cfg3_rank']['frac'], d['end_to_end'].get('ms_per_step'))
PY
