#!/usr/bin/env python3
"""Print per-kernel averages of every PMC counter in a rocprofv3 rocpd database (run on the GPU box: the
databases are too large to travel)."""
import sqlite3
import sys

for db in sys.argv[1:]:
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                       "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    for name, ctr, avg, n in rows:
        print(f"{name[:90]}|{ctr}|{avg:.1f}|{n}")
