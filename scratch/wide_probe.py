import sys, json
sys.path.insert(0, '.')
import torch
from bench import HotPathStep, time_kernel
from dpr_scale_amd import _lib
dev = torch.device('cuda', 0)
for name, opts in (("wide", {}), ("wide_nocopy", {"wide_nocopy": 1}), ("register_staged", {"no_wide": 1})):
    for k, v in opts.items():
        _lib.set_option(k, v)
    hp = HotPathStep(128, 8, 30528, 1.0, 1, 0, dev)
    us = time_kernel(hp, hp.k_sim32, reps=10, iters=5)
    print(json.dumps({"variant": name, "sim_stats_f32_us": round(us, 2)}), flush=True)
    for k in opts:
        _lib.set_option(k, 0)
    del hp
    torch.cuda.empty_cache()
