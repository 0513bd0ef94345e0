#!/usr/bin/env python3
"""Timeline of a replayed step from a rocprofv3 --kernel-trace database: per kernel of the step its average duration, the gap to
the kernel before it (end -> start) and the step's period (start of the first kernel -> start of the next step's first kernel).
    rocprofv3 --kernel-trace -d <dir> -o p -- python scripts/bench_rankstep.py --shapes 128:8:768:8 --reps 30
    python scripts/step_gaps.py <dir>/.../p_results.db
Kernel durations alone do not add up to the step: what the stream pays per kernel is duration + gap."""
import sqlite3
import sys
from collections import defaultdict


def main():
    con = sqlite3.connect(sys.argv[1])
    rows = [(s, e, n.split("(")[0].replace("void ", "").replace("dprhot::", "")) for s, e, n in
            con.execute("select start, end, name from kernels order by start") if "dprhot" in n]
    if not rows:
        print("no library kernels in the trace")
        return
    first = rows[-1][2] if False else None
    # the step's first kernel: the sim launch
    names = [r[2] for r in rows]
    head = next(n for n in names if "sim" in n)
    idx = [i for i, n in enumerate(names) if n == head]
    seqs = defaultdict(list)
    for a, b in zip(idx[:-1], idx[1:]):
        seqs[tuple(names[a:b])].append((a, b))
    seq, spans = max(seqs.items(), key=lambda kv: len(kv[1]))
    spans = spans[len(spans) // 4:]  # (drop the warm-up quarter)
    n = len(spans)
    period = sum(rows[b][0] - rows[a][0] for a, b in spans) / n / 1e3
    print(f"{n} steps of {len(seq)} kernels, period {period:.2f} us")
    tot_d = tot_g = 0.0
    for j, name in enumerate(seq):
        d = sum(rows[a + j][1] - rows[a + j][0] for a, _ in spans) / n / 1e3
        g = sum(rows[a + j][0] - rows[a + j - 1][1] for a, _ in spans if a + j - 1 >= 0) / n / 1e3
        tot_d += d
        tot_g += g
        print(f"   {name[:56]:56s} duration {d:6.2f} us   gap before it {g:6.2f} us")
    print(f"   sum of durations {tot_d:.2f} us + sum of gaps {tot_g:.2f} us")


if __name__ == "__main__":
    main()
