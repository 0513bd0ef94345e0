cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in "w8:" "w4:--opt sk_w8=0" "withg:--with-g"; do
  l=${v%%:*}; o=${v#*:}
  rm -rf /tmp/gp_$l; ( cd /tmp && timeout 200 rocprofv3 --kernel-trace -d /tmp/gp_$l -o p -- python $GRAFT_REPO_ROOT/scripts/bench_rankstep.py --shapes 128:8:768:8 --reps 30 $o ) > /tmp/gp_$l.log 2>&1
  echo "== $l: $(grep step_us /tmp/gp_$l.log | sed 's/.*"step_us": \([0-9.]*\).*/step \1 us/')"
  python scripts/step_gaps.py $(find /tmp/gp_$l -name "*.db" | head -1)
done
