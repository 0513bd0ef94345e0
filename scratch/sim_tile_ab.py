"""Round 6: the stored-logits similarity GEMM (dprhot_sim_stats, bf16 operands) of the 256-512-row band with the plan's tile against the
128 x 128 tile pinned (option tile = 0: the LDS-DMA kernel); us per launch, alternating."""
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
from bench import HotPathStep, time_kernel
from dpr_scale_amd import _lib
dev = torch.device("cuda", 0)
for B, Nc in ((256, 8192), (384, 8192), (512, 8192), (512, 16384), (768, 8192)):
    hp = HotPathStep(B, Nc // B, 768, 1.0, 1, 0, dev)
    hp.k_prep()
    r = {"B": B, "Nc": Nc, "nl": hp.nl, "plan_us": [], "tile0_us": [], "fwd_plan_us": [], "fwd_tile0_us": []}
    if hp.nl:
        print(json.dumps(r)); continue
    for rnd in range(3):
        _lib.set_option("tile", -1)
        r["plan_us"].append(round(time_kernel(hp, hp.k_sim, reps=20, iters=5), 2))
        r["fwd_plan_us"].append(round(time_kernel(hp, hp.k_fwd, reps=20, iters=5), 2))
        _lib.set_option("tile", 0)
        r["tile0_us"].append(round(time_kernel(hp, hp.k_sim, reps=20, iters=5), 2))
        r["fwd_tile0_us"].append(round(time_kernel(hp, hp.k_fwd, reps=20, iters=5), 2))
    _lib.set_option("tile", -1)
    print(json.dumps(r), flush=True)
    del hp; torch.cuda.empty_cache()
