"""A/B of the option `nt_stores` (non-temporal stores for the large fp32 outputs: 1 = default, 0 = plain stores) on the router-width
step (bench.roofline_router) and on the 8192^2 x 768 launches (bench.roofline_at_scale)."""
import json
import sys

sys.path.insert(0, '.')
import torch
import bench_blocks as bench
from dpr_scale_amd import _lib

dev = torch.device('cuda', 0)
for rnd in range(2):
    for v in (1, 0):
        _lib.set_option("nt_stores", v)
        r = bench.roofline_router(dev)
        print(json.dumps({"nt_stores": v, "router": {"sim_us": r["sim_stats_f32"]["us"], "finish_us": r["softmax_finish"]["us"],
                          "bwd_us": r["bwd_pair"]["us"], "step_us": r["step_us"], "frac": r["frac"]}}), flush=True)
        s = bench.roofline_at_scale(dev, 768)
        print(json.dumps({"nt_stores": v, "at_scale": {k: (x.get("us"), x.get("frac")) for k, x in s.items() if isinstance(x, dict) and "us" in x}}),
              flush=True)
_lib.set_option("nt_stores", 1)
