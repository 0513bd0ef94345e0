cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for m in 0 1; do
  rm -rf /tmp/pa$m
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pa$m -o p -- python $GRAFT_REPO_ROOT/scripts/bench_rankstep.py --shapes 128:8:768:8 --eager --reps 50 --opt sk_dq_atomic=$m ) > gpurun_out/pa$m.log 2>&1
  echo "== sk_dq_atomic=$m"; tail -n 1 gpurun_out/pa$m.log | cut -c1-200
  python scripts/prof_summary.py r06_atomic$m --trace "$(find /tmp/pa$m -name '*.db' | head -1)" --out gpurun_out/pa_sum | grep dprhot | cut -d, -f1-6 | cut -c1-200
done
