#!/bin/bash
# rocprofv3 kernel-trace of scripts/bench_rankstep.py under a list of option settings; prints the average duration of the step's kernels.
#   bash scripts/prof_kernels.sh <shape B:K:d:W> <label=opts ...>     opts: comma-separated name:value, "withg", "none", or lib:<path of another build of the library>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
shape=$1; shift
for item in "$@"; do
  label="${item%%=*}"; opts="${item#*=}"; args=""; unset DPRHOT_LIB
  IFS=',' read -ra parts <<< "$opts"
  for o in "${parts[@]}"; do
    case "$o" in withg) args="$args --with-g";; none|"") ;; lib:*) export DPRHOT_LIB="$GRAFT_REPO_ROOT/${o#lib:}";; *) args="$args --opt ${o/:/=}";; esac
  done
  rm -rf /tmp/pk_$label
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pk_$label -o p -- python $GRAFT_REPO_ROOT/scripts/bench_rankstep.py --shapes $shape --eager --reps 30 $args ) > /tmp/pk_$label.log 2>&1
  db=$(find /tmp/pk_$label -name "*.db" | head -1)
  echo "== $label ($args): $(grep step_us /tmp/pk_$label.log | sed 's/.*"step_us": \([0-9.]*\).*/step \1 us/')"
  python - "$db" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
for name, n, avg, mn in con.execute("select name, count(*), avg(duration), min(duration) from kernels group by name order by sum(duration) desc limit 5"):
    if "dprhot" in name and n > 100:
        print(f"   {name.split('(')[0][:60]:60s} n={n} avg {avg/1e3:.2f} us min {mn/1e3:.2f}")
PY
done
