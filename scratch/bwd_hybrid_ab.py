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
for B, Nc, d in ((1024, 8192, 768), (1024, 16384, 768), (1024, 32768, 768), (1024, 49152, 768), (512, 16384, 768), (512, 32768, 768), (2048, 16384, 768), (2048, 32768, 768), (2048, 65536, 768), (1024, 32768, 1024), (512, 8192, 768)):
    G = (torch.randn(B, Nc, device=dev) * 0.01).to(torch.bfloat16); Qb = torch.randn(B, d, device=dev).to(torch.bfloat16); Cb = torch.randn(Nc, d, device=dev).to(torch.bfloat16)
    go = torch.ones(1, device=dev)
    r = {"B": B, "Nc": Nc, "d": d}
    for rnd in range(2):
        _lib.set_option("sk_dbg", 1000 + 4096); r.setdefault("pair_us", []).append(round(t(lambda: kn.inbatch_bwd(G, Qb, Cb, 1.0, go)), 1))
        _lib.set_option("sk_dbg", 1000 + 1); r.setdefault("dc_apart_us", []).append(round(t(lambda: kn.inbatch_bwd(G, Qb, Cb, 1.0, go)), 1))
    _lib.set_option("sk_dbg", 0)
    print(json.dumps(r), flush=True)
