"""Host time of one eager cfg2 step: N calls issued behind a synchronize, clock stopped BEFORE the device is waited for (the queue holds them all).
Prints the per-call host cost next to the device-paced loop's time per step, for the C call alone and for its two launches replaced by
torch's cheapest kernels (the HIP launch path without the library)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import HotPathStep  # noqa: E402

dev = torch.device("cuda", 0)
hp = HotPathStep(32, 8, 768, 1.0, 1, 0, dev)
for _ in range(200):
    hp.c_step()
torch.cuda.synchronize()
out = {}
for n in (50, 200, 1000):
    best = 1e9
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter_ns()
        for _ in range(n):
            hp.c_step()
        t1 = time.perf_counter_ns()
        torch.cuda.synchronize()
        t2 = time.perf_counter_ns()
        best = min(best, (t1 - t0) / n)
        last = ((t1 - t0) / n, (t2 - t0) / n)
    out[f"n={n}"] = {"host_issue_us_per_call_best": round(best / 1e3, 2), "last_issue_us": round(last[0] / 1e3, 2), "last_total_us": round(last[1] / 1e3, 2)}
x = torch.zeros(64, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter_ns()
for _ in range(400):
    x.add_(1.0)
t1 = time.perf_counter_ns()
torch.cuda.synchronize()
out["torch_add__issue_us_per_launch"] = round((t1 - t0) / 400 / 1e3, 2)
print(json.dumps(out))
