#!/usr/bin/env python3
"""Per-rank step of a multi-rank configuration timed on ONE GPU: the gathered packed buffer is filled locally (every
rank slot gets this rank's rows -- timing only, parity lives in tests/), then dprhot_inbatch_step_packed_f32 is replayed.
  python scripts/bench_rankstep.py [--shapes B:K:d:W,...] [--eager]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import HotPathStep, time_kernel  # noqa: E402
from dpr_scale_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="128:8:768:8,64:2:1024:8,32:8:768:8")
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--opt", action="append", default=[], help="name=value: dprhot_set_option before anything runs (A/B of the plans)")
    ap.add_argument("--with-g", action="store_true", help="ask for the dScores (G != NULL): the plan that materialises them")
    a = ap.parse_args()
    if a.with_g:
        os.environ["DPRHOT_BENCH_G"] = "1"
    for kv in a.opt:
        k, v = kv.split("=")
        _lib.set_option(k, int(v))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    for sh in a.shapes.split(","):
        B, K, d, W = (int(x) for x in sh.split(":"))
        hp = HotPathStep(B, K, d, 1.0, W, 0, dev, dist_mode=True)
        hp.k_pack()
        for r in range(W):
            hp.Cb[r * hp.rows_c:(r + 1) * hp.rows_c].copy_(hp.send)
        torch.cuda.synchronize()
        us = time_kernel(hp, hp.k_step, reps=a.reps, iters=10, use_graph=not a.eager)
        bn, bd, nd = float(B) * hp.Nc, float(B) * d, float(hp.Nc) * d
        algo = (4 * bd + 2 * nd + 4 * bn) + 6 * bn + (2 * bn + 2 * nd + 4 * bd) + (2 * bn + 2 * bd + 4 * nd)  # SURVEY 8(d), 3-kernel design
        print(json.dumps({"B": B, "K": K, "d": d, "W": W, "Nc": hp.Nc, "step_us": round(us, 2), "loss_sum": float(hp.loss_sum.item()),
                          "algorithmic_MB": round(algo / 1e6, 2), "hbm_frac": round(algo / us * 1e-3 / 8000.0, 4),
                          "few_rows_plan_enabled": not _lib.get_option("no_skinny"), "fused_forward": ("logits stored", "one pass 256x256", "one pass 128x128")[_lib.fwd_one_pass(hp.B, hp.Nc, hp.d)] if hp.want_g else "few-rows plan (no dScores)",
                          "G_materialised": bool(hp.want_g)}), flush=True)
        del hp
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
