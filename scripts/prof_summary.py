#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases (ROCm 7.2 default output) into small CSV/markdown files under profiles/.

  python scripts/prof_summary.py <tag> [--trace gpurun_out/prof_trace/bench_results.db]
                                        [--fetch gpurun_out/prof_fetch/bench_results.db]
                                        [--write gpurun_out/prof_write/bench_results.db]
Kernel-trace stats: calls, total, average, min, max duration per kernel (the `--stats` table).
PMC: per-kernel average FETCH_SIZE / WRITE_SIZE (KiB) and the HBM bytes derived from them with the gfx950
correction of MI355X_MICROARCH.md (FETCH_SIZE reports 1/2 of wide coalesced reads -> x2; WRITE_SIZE as is).
"""
import argparse
import csv
import os
import sqlite3


def q(db, sql):
    con = sqlite3.connect(db)
    try:
        return con.execute(sql).fetchall()
    finally:
        con.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--trace")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--out", default="profiles")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    rows = {}
    if a.trace:
        for name, n, tot, avg, mn, mx in q(a.trace, "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                                                    "from kernels group by name order by sum(duration) desc"):
            rows[name] = dict(kernel=name, calls=n, total_us=tot / 1e3, avg_us=avg / 1e3, min_us=mn / 1e3, max_us=mx / 1e3)
        tot = sum(r["total_us"] for r in rows.values()) or 1.0
        for r in rows.values():
            r["pct"] = 100.0 * r["total_us"] / tot
    for key, db in (("FETCH_SIZE", a.fetch), ("WRITE_SIZE", a.write)):
        if not db:
            continue
        for name, avg in q(db, f"select kernel_name, avg(value) from counters_collection where counter_name='{key}' group by kernel_name"):
            rows.setdefault(name, dict(kernel=name))[key + "_KiB_avg"] = avg
    for r in rows.values():
        f, w = r.get("FETCH_SIZE_KiB_avg"), r.get("WRITE_SIZE_KiB_avg")
        if f is not None and w is not None:
            r["hbm_bytes_per_launch_corrected"] = (2.0 * f + w) * 1024.0
    cols = ["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "FETCH_SIZE_KiB_avg", "WRITE_SIZE_KiB_avg",
            "hbm_bytes_per_launch_corrected"]
    path = os.path.join(a.out, f"{a.tag}_kernel_stats.csv")
    with open(path, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=cols)
        w.writeheader()
        for r in sorted(rows.values(), key=lambda r: -r.get("total_us", 0)):
            w.writerow({k: (f"{v:.3f}" if isinstance(v, float) else v) for k, v in r.items()})
    print(open(path).read())
    # per-launch HBM traffic in the names bench.py uses (read back by bench.py -> roofline.traffic / roofline_cfg3_rank.traffic)
    import json
    method = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB per "
              "MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts 64 B per 128 B request)")

    def entry(r):
        return {"hbm_bytes_per_launch": round(r["hbm_bytes_per_launch_corrected"]), "FETCH_SIZE_KiB": round(r["FETCH_SIZE_KiB_avg"], 2),
                "WRITE_SIZE_KiB": round(r["WRITE_SIZE_KiB_avg"], 2), "rocprof_avg_us": round(r.get("avg_us", 0.0), 3),
                "calls": r.get("calls"), "kernel": r["kernel"][:120]}

    if "cfg3rank" in a.tag:
        # every library kernel launched once per step of scripts/bench_rankstep.py (set-up launches run once and are left out)
        lib = [r for r in rows.values() if "dprhot::" in r["kernel"] and r.get("hbm_bytes_per_launch_corrected") is not None and r.get("calls")]
        steps = max((r["calls"] for r in lib), default=0)
        per_step = [r for r in lib if r["calls"] >= steps // 2]
        if per_step:
            kern = {r["kernel"].split("dprhot::")[1].split("(")[0].split("<")[0]: entry(r) for r in per_step}
            total = sum(r["hbm_bytes_per_launch_corrected"] * r["calls"] for r in per_step) / steps
            jp = os.path.join(a.out, "cfg3rank_traffic.json")
            json.dump({"source": a.tag, "method": method + " on scripts/bench_rankstep.py --shapes 128:8:768:8 --eager", "steps": steps,
                       "step_hbm_bytes": round(total), "kernels": kern}, open(jp, "w"), indent=1)
            print("wrote", jp)
    else:
        names = {"sim_stats_f32": "EpiSim", "softmax_finish": "gfinal", "bwd_pair": "gemm_pair_kernel", "softmax_bwd_fused": "step_small_kernel"}
        traffic = {}
        for bname, pat in names.items():
            for r in rows.values():
                if pat in r["kernel"] and r.get("hbm_bytes_per_launch_corrected") is not None:
                    traffic[bname] = entry(r)
        if traffic:
            jp = os.path.join(a.out, "bench_cfg2_traffic.json")
            json.dump({"source": a.tag, "method": method + " on `python bench.py`", "kernels": traffic}, open(jp, "w"), indent=1)
            print("wrote", jp)

if __name__ == "__main__":
    main()
