#!/usr/bin/env python3
"""Roofline sweep of the hot-path launches over (B, Nc) -- the BASELINE shapes are launch/latency-bound, so this
shows where each kernel goes once the problem is large enough to be bandwidth- or MFMA-bound.

  python bench_sweep.py [--dim 768] [--shapes 32x256,128x8192,...]
Prints one JSON line per shape: per-launch average duration (HIP events, 20-launch graphs replayed back to back),
algorithmic GB/s and TFLOP/s and their fractions of the gfx950 peaks.
"""
import argparse
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

from bench import HBM_PEAK_GBS, MFMA_PEAK_TFLOPS, HotPathStep, time_kernel  # noqa: E402

DEFAULT = "32x256,32x2048,8x512,64x1024,128x8192,256x8192,1024x8192,4096x8192,8192x8192,1024x65536,8192x65536"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--shapes", default=DEFAULT)
    ap.add_argument("--opt", default="", help="comma list of library options name=value (dprhot_set_option) applied before the sweep, e.g. big_min=64")
    a = ap.parse_args()
    from dpr_scale_amd import _lib
    if a.opt:
        for kv in a.opt.split(","):
            k, v = kv.split("=")
            _lib.set_option(k, int(v))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    for sh in a.shapes.split(","):
        B, Nc = (int(x) for x in sh.split("x"))
        d = a.dim
        K = Nc // B
        hp = HotPathStep(B, K, d, 1.0, 1, 0, dev)
        bn, bd, nd = float(B) * Nc, float(B) * d, float(Nc) * d
        kern = {
            "sim_stats_f32": (hp.k_sim32, 6 * bd + 6 * nd + 4 * bn, 2 * bn * d),
            "prep": (hp.k_prep, 6 * (bd + nd), 0.0),
            "sim_stats_bf16": (hp.k_sim, 2 * (bd + nd) + (0 if hp.nl else 4 * bn), 2 * bn * d),
            "softmax_finish": (hp.k_softmax, 8 * bn / 64 if hp.nl else 6 * bn, 0.0),
        }
        if hp.nl:  # no-logits forward: the dScores come from a GEMM pass that recomputes the logits (two-pass form), or the whole forward
            kern["dscores"] = (hp.k_dscores, 2 * (bd + nd) + 2 * bn, 2 * bn * d)   # is ONE GEMM pass + a row kernel (fwd_bf16)
        kern["fwd_bf16"] = (hp.k_fwd, 2 * (bd + nd) + (4 * bn if (hp.nl or _lib.fwd_one_pass(B, Nc, d)) else 10 * bn), 2 * bn * d)   # dprhot_inbatch_fwd: everything up to G and the loss
        kern["bwd_pair"] = (hp.k_bwd, 4 * bn + 6 * (bd + nd), 4 * bn * d)
        # THE STEP: one call of dprhot_inbatch_step_f32 (fp32 embeddings in, dQ / dC_part out) -- what a training iteration runs.  (Until
        # round 5 `step_us` was the SUM of the rows above, which counts the similarity GEMM twice -- sim_stats_f32 and prep + sim_stats_bf16
        # are alternatives, a step runs one of them: 1024 x 8192 read 133 us where the step takes ~95.)
        kern["step"] = (hp.k_step, (4 * bd + 4 * nd) + (2 * (bd + nd) + 4 * bn) + 4 * bn + 6 * (bd + nd), 6 * bn * d)
        reps = 20 if bn * d < 1e11 else 4
        # forward_plan: what the separate calls do (sim_stats / softmax_finish rows); fused_forward: how fwd_bf16 and the step form G
        # (dprhot_fwd_one_pass: 0 logits stored, 1 one pass on the 256 x 256 tile, 2 one pass on the 128 x 128 LDS-DMA tile)
        one = _lib.fwd_one_pass(B, Nc, d)
        row = {"B": B, "Nc": Nc, "d": d, "forward_plan": "no-logits" if hp.nl else "logits",
               "fused_forward": ("logits stored", "one pass, 256 x 256 tile", "one pass, 128 x 128 tile")[one]}
        for name, (fn, by, fl) in kern.items():
            us = time_kernel(hp, fn, reps=reps, iters=5)
            row[name] = {"us": round(us, 2), "GBps": round(by / us * 1e-3, 1), "hbm_frac": round(by / us * 1e-3 / HBM_PEAK_GBS, 4),
                         "TFLOPs": round(fl / us * 1e-6, 2), "mfma_frac": round(fl / us * 1e-6 / MFMA_PEAK_TFLOPS, 4)}
        row["step_us"] = row["step"]["us"]
        row["hbm_floor_us"] = round(max(kern["step"][1] / (HBM_PEAK_GBS * 1e3), kern["step"][2] / (MFMA_PEAK_TFLOPS * 1e6)), 2)
        row["pairs_per_s"] = round(B / row["step_us"] * 1e6, 1)
        print(json.dumps(row), flush=True)
        del hp
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
