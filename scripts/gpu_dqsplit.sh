cd $GRAFT_REPO_ROOT
for s in 1 2 3 4 9; do echo "== DPRHOT_DQ_SPLITS=$s"; DPRHOT_DQ_SPLITS=$s timeout 300 ./dpr_scale_amd/selftest time 2>&1 | grep -E "^case|TIME (dq |inbatch_bwd)|FAIL" | grep -A2 -E "B=32 Nc=2112|B=32 Nc=256|B=8 Nc=512|B=64 Nc=1024" | grep -v "^--"; done
