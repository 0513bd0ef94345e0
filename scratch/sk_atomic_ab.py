"""A/B of option sk_dq_atomic (round 6) on the per-rank steps of the BASELINE multi-rank configurations, one process, the two forms
alternating over rounds; each sample = a HIP graph of 20 steps replayed 10 times (us per step)."""
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from bench import HotPathStep, time_kernel  # noqa: E402
from dpr_scale_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
for sh in os.environ.get("AB_SHAPES", "128:8:768:8,128:8:768:4,128:8:1024:8,64:8:768:8").split(","):
    B, K, d, W = (int(x) for x in sh.split(":"))
    hp = HotPathStep(B, K, d, 1.0, W, 0, dev, dist_mode=True)
    hp.k_pack()
    for r in range(W):
        hp.Cb[r * hp.rows_c:(r + 1) * hp.rows_c].copy_(hp.send)
    torch.cuda.synchronize()
    res = {0: [], 1: []}
    for rnd in range(4):
        for mode in (0, 1):
            _lib.set_option("sk_dq_atomic", mode)
            res[mode].append(round(time_kernel(hp, hp.k_step, reps=20, iters=10), 2))
    _lib.set_option("sk_dq_atomic", 0)
    print(json.dumps({"B": B, "K": K, "d": d, "W": W, "Nc": hp.Nc, "G_materialised": bool(hp.want_g), "slabs_us": res[0], "atomic_us": res[1]}), flush=True)
    del hp
    torch.cuda.empty_cache()
