#!/usr/bin/env python3
"""Reads the rocprofv3 database of scripts/overlap_trace.py (kernel trace + memory-copy trace) and checks, for the LAST traced training
step, what SURVEY.md section 8(e) / north_star ask of the collectives:
  (i)  the all-gather of the packed context rows runs on another stream than the compute stream, inside the window
       [end of dprhot::pack_ctx_kernel, start of the step's first similarity kernel] -- i.e. under the query tower's forward -- and the
       compute stream is busy with tower kernels while it runs;
  (ii) the reduce-scatter of the dC partials runs on another stream inside [end of dprhot::rescale_grads_kernel (the operator's
       backward), start of the launch that consumes its result (dprhot::grad_unpack_kernel, the widen of the bf16 wire)] -- i.e.
       under the query tower's backward -- again with the compute stream busy.
Exit code 0 and a JSON summary when both hold.  On a one-rank world RCCL may carry a collective as a device-to-device copy instead
of an ncclDevKernel: both kinds of operation are looked at.

    python scripts/overlap_check.py <results.db> [--out profiles/r04_overlap_summary.json] [--dump]
"""
import argparse
import json
import sqlite3
import sys


def rows(con, sql):
    cur = con.execute(sql)
    names = [d[0] for d in cur.description]
    return [dict(zip(names, r)) for r in cur.fetchall()]


def pick(d, *cands):
    for c in cands:
        if c in d and d[c] is not None:
            return d[c]
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--out")
    ap.add_argument("--dump", action="store_true")
    ap.add_argument("--timeline", help="write the operations around the two collectives to this text file")
    ap.add_argument("--require-busy", type=float, default=0.0, help="also require the compute stream to be busy for at least this fraction of each "
                                                                   "collective's duration, with >= 10 tower kernels that START AND END inside it "
                                                                   "(a collective that takes time: overlap_trace.py --pad-mb)")
    a = ap.parse_args()
    con = sqlite3.connect(a.db)
    objs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    if a.dump:
        for o in objs:
            cols = [r[1] for r in con.execute(f"pragma table_info('{o}')")]
            print(o, cols)
    kview = "kernels" if "kernels" in objs else next(o for o in objs if "kernel" in o.lower())
    ks = rows(con, f"select * from {kview}")
    ops = []
    for k in ks:
        ops.append({"kind": "kernel", "name": k["name"], "start": pick(k, "start", "start_timestamp"), "end": pick(k, "end", "end_timestamp"),
                    "stream": pick(k, "stream_id", "stream", "queue_id", "queue")})
    cview = next((o for o in objs if o.lower() in ("memory_copies", "memory_copy")), None) or next((o for o in objs if "memory_cop" in o.lower()), None)
    if cview:
        for c in rows(con, f"select * from {cview}"):
            ops.append({"kind": "copy", "name": str(pick(c, "name", "kind", "direction")), "start": pick(c, "start", "start_timestamp"),
                        "end": pick(c, "end", "end_timestamp"), "stream": pick(c, "stream_id", "stream", "queue_id", "queue"),
                        "bytes": pick(c, "size", "bytes")})
    ops = [o for o in ops if o["start"] is not None and o["end"] is not None]
    ops.sort(key=lambda o: o["start"])
    if a.dump:
        for o in ops[-60:]:
            print(o)

    def last(pred, before=None):
        c = [o for o in ops if pred(o) and (before is None or o["start"] < before)]
        return c[-1] if c else None

    def first(pred, after):
        c = [o for o in ops if pred(o) and o["start"] >= after]
        return c[0] if c else None

    is_k = lambda o, s: o["kind"] == "kernel" and s in o["name"]
    # the last step that was traced completely: last rescale_grads with a pack_ctx before it
    resc = last(lambda o: is_k(o, "rescale_grads_kernel"))
    assert resc is not None, "no dprhot::rescale_grads_kernel in the trace (did the step take the multi-rank branch?)"
    pack = last(lambda o: is_k(o, "pack_ctx_kernel"), before=resc["start"])
    assert pack is not None, "no dprhot::pack_ctx_kernel in front of the backward"
    main_stream = pack["stream"]
    hot = first(lambda o: o["kind"] == "kernel" and "dprhot::" in o["name"] and "pack_ctx" not in o["name"], pack["end"])
    widen = first(lambda o: is_k(o, "grad_unpack_kernel"), resc["end"])
    assert hot is not None and widen is not None, "landmarks missing (first similarity kernel / widen launch)"

    timeline = []

    def busy(lo, hi):  # time the compute stream spends in kernels inside [lo, hi]
        return sum(max(0, min(o["end"], hi) - max(o["start"], lo)) for o in ops if o["kind"] == "kernel" and o["stream"] == main_stream)

    def window(name, lo, hi):
        cand = [o for o in ops if o["start"] >= lo and o["end"] <= hi and o["stream"] != main_stream
                and (o["kind"] == "copy" or any(t in o["name"].lower() for t in ("nccl", "rccl", "copybuffer", "allgather", "reducescatter")))]
        assert cand, f"{name}: no collective operation on another stream inside the window"
        op = max(cand, key=lambda o: o["end"] - o["start"])
        dur = op["end"] - op["start"]
        b = busy(op["start"], op["end"])
        b_after = busy(op["end"], op["end"] + dur)  # the same span right behind it: how busy the tower keeps the stream on its own
        tower = [o for o in ops if o["kind"] == "kernel" and o["stream"] == main_stream and lo <= o["start"] <= hi]
        before = sum(1 for o in tower if o["end"] <= op["start"])
        after = sum(1 for o in tower if o["start"] >= op["end"])
        inside = sum(1 for o in tower if o["start"] >= op["start"] and o["end"] <= op["end"])
        # does the compute stream wait for it?  the gap between the two tower kernels around the operation against the window's median gap
        gaps = sorted(b2["start"] - a2["end"] for a2, b2 in zip(tower, tower[1:]))
        prev = [o for o in tower if o["start"] <= op["start"]]
        nxt = [o for o in tower if o["start"] > op["start"]]
        gap_here = (nxt[0]["start"] - prev[-1]["end"]) if prev and nxt else None
        timeline.append((name, [o for o in ops if op["start"] - 60e3 <= o["start"] <= op["end"] + 60e3]))
        return {"operation": op["name"][:120], "kind": op["kind"], "stream": op["stream"], "compute_stream": main_stream, "duration_us": round(dur / 1e3, 2),
                "window_us": round((hi - lo) / 1e3, 1), "offset_in_window_us": round((op["start"] - lo) / 1e3, 2),
                "tower_kernels_in_window": len(tower), "tower_kernels_before_it": before, "tower_kernels_after_it": after,
                "tower_kernels_starting_and_ending_inside_it": inside,
                "compute_stream_busy_frac_during_it": round(b / max(dur, 1), 3),
                "compute_stream_busy_frac_in_the_same_span_right_after_it": round(b_after / max(dur, 1), 3),
                "compute_stream_gap_around_it_us": None if gap_here is None else round(gap_here / 1e3, 2),
                "median_gap_between_tower_kernels_us": round(gaps[len(gaps) // 2] / 1e3, 2) if gaps else None, "bytes": op.get("bytes")}

    out = {"all_gather_under_query_tower_forward": window("all-gather", pack["end"], hot["start"]),
           "reduce_scatter_under_query_tower_backward": window("reduce-scatter", resc["end"], widen["start"]),
           "landmarks": {"window_i": "dprhot::pack_ctx_kernel end -> " + hot["name"][:60] + " start",
                         "window_ii": "dprhot::rescale_grads_kernel end -> dprhot::grad_unpack_kernel (widen) start"}}
    # shown = the operation runs on ANOTHER stream, starts behind its producer and is finished before its consumer's window ends, with
    # the tower's kernels (>= 10 of them) still to come on the compute stream when it ends: the tower does not queue behind it, and
    # the consumer -- the wait is right in front of it -- finds it done.  (On a one-rank world the "collective" is a 4-5 us device copy:
    # it is over before the host has even launched the tower's first kernel, so "busy while it runs" is reported, not required.  A
    # W > 1 timeline, where the transfer takes tens of microseconds, is what a multi-GPU box will add.)
    ok = all(v["stream"] != v["compute_stream"] and v["tower_kernels_after_it"] >= 10 and v["offset_in_window_us"] >= 0
             and v["offset_in_window_us"] + v["duration_us"] < v["window_us"]
             for k, v in out.items() if k != "landmarks")
    if a.require_busy > 0:
        # busy for the required fraction of the collective -- or as busy as the tower keeps the stream by itself right behind it (the
        # towers have host-side gaps of their own: a collective cannot be blamed for those) -- with tower kernels that start AND end
        # inside the collective's interval
        ok = ok and all((v["compute_stream_busy_frac_during_it"] >= a.require_busy or
                         v["compute_stream_busy_frac_during_it"] >= 0.9 * v["compute_stream_busy_frac_in_the_same_span_right_after_it"])
                        and v["tower_kernels_starting_and_ending_inside_it"] >= 4 for k, v in out.items() if k != "landmarks")
        out["required_busy_frac"] = a.require_busy
    out["overlap_shown"] = bool(ok)
    print(json.dumps(out, indent=1))
    if a.timeline:
        with open(a.timeline, "w") as fh:
            for name, tl in timeline:
                fh.write(f"# {name}: operations within 60 us of it (start_us relative to the first listed, duration_us, stream, kind, name)\n")
                t0 = tl[0]["start"] if tl else 0
                for o in tl:
                    fh.write(f"{(o['start'] - t0) / 1e3:9.2f} {(o['end'] - o['start']) / 1e3:8.2f}  stream {o['stream']}  {o['kind']:6s} {o['name'][:90]}\n")
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
