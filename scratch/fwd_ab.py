"""dprhot_inbatch_fwd_f32 + dprhot_inbatch_bwd as separate calls (the flow of a subclass with its own loss on sim_score's logits): the few-rows
plan's launches (default) against the generic kernels (option no_skinny), HIP-event time per call over a graph."""
import json
import sys

sys.path.insert(0, '.')
import torch
from bench import HotPathStep, time_kernel
from dpr_scale_amd import _lib

dev = torch.device('cuda', 0)
for B, K in ((128, 64), (128, 8), (64, 16), (32, 64)):
    row = {"B": B, "Nc": B * K, "d": 768}
    for name, off in (("few_rows_plan", 0), ("generic", 1)):
        _lib.set_option("no_skinny", off)
        hp = HotPathStep(B, K, 768, 1.0, 1, 0, dev)
        row[name] = {"fwd_f32_us": round(time_kernel(hp, hp.k_fwd32, reps=20, iters=10), 2), "bwd_us": round(time_kernel(hp, hp.k_bwd, reps=20, iters=10), 2)}
        del hp
    _lib.set_option("no_skinny", 0)
    print(json.dumps(row), flush=True)
