cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "DPRHOT_SK_KSTEPS=13" "DPRHOT_SK_KSTEPS=17" "DPRHOT_SK_KSTEPS=22" "DPRHOT_SK_KSTEPS=26" "DPRHOT_SK_KSTEPS=33" "DPRHOT_SK_KSTEPS=44"; do
  echo "== $v"; ( env $v timeout 300 python scripts/bench_rankstep.py --shapes 128:8:768:8,64:2:1024:8 ) 2>&1 | grep "^{" | cut -c1-110
done
