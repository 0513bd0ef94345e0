#!/usr/bin/env python3
"""A/B of option g8_one_tile at 8192^2 x 768 (dScores pass, stored logits): persistent workgroups against one workgroup per tile,
alternating in one process.  python scratch/g8_one_tile_ab.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import roofline_at_scale  # noqa: E402
from dpr_scale_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for rnd in range(3):
    for v in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '0,1').split(',')]:
        _lib.set_option("g8_one_tile", v)
        r = roofline_at_scale(dev, 768)
        print(json.dumps({"g8_one_tile": v, "dscores_gemm_us": r["dscores_gemm"]["us"], "sim_store_us": r["sim_store"]["us"], "sim_gemm_us": r["sim_gemm"]["us"]}), flush=True)
_lib.set_option("g8_one_tile", 0)
