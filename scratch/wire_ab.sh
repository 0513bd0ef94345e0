cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for k in 2 0 2 0; do timeout 100 python scratch/wire_ab.py --kind $k | tail -1; done
for k in 2 0; do
  rm -rf /tmp/wab$k; ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/wab$k -o p -- python $GRAFT_REPO_ROOT/scratch/wire_ab.py --kind $k --eager ) > /tmp/wab$k.log 2>&1
  db=$(find /tmp/wab$k -name "*.db" | head -1)
  python - "$db" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
for name, n, avg, mn in con.execute("select name, count(*), avg(duration), min(duration) from kernels group by name order by sum(duration) desc limit 6"):
    if n > 100: print(f"   {name.split('(')[0][:60]:60s} n={n} avg {avg/1e3:.2f} us min {mn/1e3:.2f}")
PY
done
