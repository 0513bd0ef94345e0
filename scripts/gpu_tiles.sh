cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for t in 0 1 2 3 4 5; do ( DPRHOT_TILE=$t timeout 300 ./dpr_scale_amd/selftest time ) > gpurun_out/tiles_$t.log 2>&1; echo "tile$t rc=$?"; done
