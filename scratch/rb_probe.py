"""Step time (HIP events, graph of 20 steps) with the fused softmax + backward kernel (two launches) and without it (three), at the
shapes the two-row-block variant (32 < B <= 64) newly covers, and at cfg2 as the unchanged reference."""
import json
import sys

sys.path.insert(0, '.')
import torch
from bench import HotPathStep, time_kernel
from dpr_scale_amd import _lib

dev = torch.device('cuda', 0)
for B, K, d in ((32, 8, 768), (64, 8, 768), (64, 2, 1024), (64, 4, 768), (48, 12, 768), (64, 12, 768)):
    res = {"B": B, "K": K, "d": d}
    for name, off in (("fused_2_launches_us", 0), ("three_launches_us", 1)):
        _lib.set_option("no_small_step", off)
        hp = HotPathStep(B, K, d, 1.0, 1, 0, dev)
        res[name] = round(time_kernel(hp, hp.k_step, reps=20, iters=10), 2)
        del hp
    _lib.set_option("no_small_step", 0)
    print(json.dumps(res), flush=True)
