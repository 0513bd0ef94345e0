"""dprhot_inbatch_bwd at mid-size batches: the library's plan against its alternatives (options no_big_bwd: the generic pair kernel on
128 / 64-row tiles; unfused_bwd: dQ and dC as two launches of the 128 x 128 engine), us per call, alternating in one process."""
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
from dpr_scale_amd.hotpath import HipKernels
from dpr_scale_amd import _lib
kn = HipKernels(); dev = torch.device("cuda", 0)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
shapes = [(int(a), int(b), 768) for a, b in (x.split("x") for x in (sys.argv[1] if len(sys.argv) > 1 else "256x8192,256x4096,384x8192,512x4096,512x8192,768x8192,1024x4096,1024x8192,256x16384,256x32768,384x16384").split(","))]
for B, Nc, d in shapes:
    G = (torch.randn(B, Nc, device=dev) * 0.01).to(torch.bfloat16); Qb = torch.randn(B, d, device=dev).to(torch.bfloat16); Cb = torch.randn(Nc, d, device=dev).to(torch.bfloat16)
    go = torch.ones(1, device=dev)
    r = {"B": B, "Nc": Nc, "d": d}
    for rnd in range(2):
        for name, opts in (("plan", {}), ("no_big_bwd", {"no_big_bwd": 1}), ("unfused", {"unfused_bwd": 1})):
            for k, v in opts.items(): _lib.set_option(k, v)
            try:
                r.setdefault(name + "_us", []).append(round(t(lambda: kn.inbatch_bwd(G, Qb, Cb, 1.0, go)), 1))
            except Exception as e:
                r[name + "_us"] = repr(e)[:60]
            for k in opts: _lib.set_option(k, 0)
    print(json.dumps(r), flush=True)
