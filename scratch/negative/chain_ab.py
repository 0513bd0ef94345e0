"""cfg2 step: one launch (chain.h) against two (option no_chain), HIP-event time over a graph of 20 steps and the eager C-ABI loop."""
import json
import sys
import time

sys.path.insert(0, '.')
import torch
from bench import HotPathStep, time_kernel
from dpr_scale_amd import _lib

dev = torch.device('cuda', 0)
for rnd in range(2):
    for v in (0, 1):
        _lib.set_option("no_chain", v)
        hp = HotPathStep(32, 8, 768, 1.0, 1, 0, dev)
        g = time_kernel(hp, hp.k_step, reps=30, iters=20)
        for _ in range(200):
            hp.c_step()
        torch.cuda.synchronize()
        ts = []
        for _ in range(15):
            t0 = time.perf_counter()
            for _ in range(500):
                hp.c_step()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 500 * 1e6)
        ts.sort()
        print(json.dumps({"no_chain": v, "graph20_us": round(g, 3), "eager_us_median": round(ts[len(ts) // 2], 3), "eager_us_min": round(ts[0], 3)}), flush=True)
        del hp
_lib.set_option("no_chain", 0)
