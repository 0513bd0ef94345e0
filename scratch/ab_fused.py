"""A/B of the cfg2 step on ONE box: the in-tree library against another build of it (DPRHOT_LIB), alternating processes.
    python scratch/ab_fused.py scratch/libdprhot_prenrb.so
Each child prints the step time (HIP events over a graph of 20 steps), the sim launch alone and their difference."""
import json
import os
import subprocess
import sys

CHILD = r'''
import json, sys
sys.path.insert(0, '.')
import torch
from bench import HotPathStep, time_kernel
dev = torch.device('cuda', 0)
hp = HotPathStep(32, 8, 768, 1.0, 1, 0, dev)
out = {"step_us": round(time_kernel(hp, hp.k_step, reps=30, iters=20), 3)}
out["sim_us"] = round(time_kernel(hp, hp.k_sim32, reps=30, iters=20), 3)
out["fused_us"] = round(out["step_us"] - out["sim_us"], 3)  # the second launch has no entry point of its own
print(json.dumps(out))
'''

if __name__ == "__main__":
    other = sys.argv[1]
    for rnd in range(3):
        for tag, lib in (("in-tree", None), ("other", other)):
            env = dict(os.environ)
            if lib:
                env["DPRHOT_LIB"] = os.path.abspath(lib)
            r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
            print(tag, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
