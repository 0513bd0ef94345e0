cd $GRAFT_REPO_ROOT
for cfg in "2 11" "2 22" "0 8" "0 16" "0 32" "0 43" "0 64"; do set -- $cfg; echo "== DQ_TILE=$1 SPLITS=$2"; DPRHOT_DQ_TILE=$1 DPRHOT_DQ_SPLITS=$2 timeout 300 ./dpr_scale_amd/selftest time 2>&1 | grep -E "^case|TIME (dq |inbatch_bwd)|FAIL" | grep -A2 -E "B=128 Nc=8192" | grep -v "^--"; done
