"""Router-width shape: the sim launch alone (back to back: the 141 MB of operands stay in the 256 MB Infinity Cache) and the whole
step (the backward's 141 MB of gradients push them out again) for the LDS-DMA sim (wide.h) and the register-staged one."""
import json
import sys

sys.path.insert(0, '.')
import torch
from bench import HotPathStep, time_kernel
from dpr_scale_amd import _lib

dev = torch.device('cuda', 0)
for name, opts in (("wide", {}), ("register_staged", {"no_wide": 1})):
    for k, v in opts.items():
        _lib.set_option(k, v)
    hp = HotPathStep(128, 8, 30528, 1.0, 1, 0, dev)
    sim = time_kernel(hp, hp.k_sim32, reps=10, iters=5)
    fin = time_kernel(hp, hp.k_softmax, reps=10, iters=5)
    bwd = time_kernel(hp, hp.k_bwd, reps=10, iters=5)
    step = time_kernel(hp, hp.k_step, reps=10, iters=5)
    print(json.dumps({"variant": name, "sim_alone_us": round(sim, 2), "softmax_us": round(fin, 2), "bwd_us": round(bwd, 2), "step_us": round(step, 2)}), flush=True)
    for k in opts:
        _lib.set_option(k, 0)
    del hp
    torch.cuda.empty_cache()
