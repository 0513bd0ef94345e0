cd $GRAFT_REPO_ROOT
for t in 2 1 0; do
  export DPRHOT_WIDE_TILE=$t
  echo "wide tile $t"
  timeout 300 python - <<'PY'
import json, torch, bench
r = bench.roofline_router(torch.device("cuda", 0))
print(json.dumps({k: r[k] for k in ("sim_stats_f32", "softmax_finish", "bwd_pair", "step_us", "frac")}))
PY
  timeout 300 python -m pytest tests -m gpu -q -x -k "router or 30522 or citadel" 2>&1 | tail -1
done
