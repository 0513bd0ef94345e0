"""Round 6: long context axis -- dC on the LDS-DMA 128 x 128 tile (plan) against dC tiles ALONE on the 256 x 256 phase-interleaved kernel
(option dc_alone_8p = 1); us per dprhot_inbatch_bwd call, alternating."""
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
from dpr_scale_amd import _lib
from dpr_scale_amd.hotpath import HipKernels
kn = HipKernels(); dev = torch.device("cuda", 0)
def t(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return round(e0.elapsed_time(e1) / n * 1e3, 1)
d = 768; go = torch.ones(1, device=dev)
for B, Nc in ((1024, 32768), (1024, 65536), (2048, 65536), (1024, 49152), (1536, 65536)):
    G = (torch.randn(B, Nc, device=dev) * 0.01).to(torch.bfloat16); Qb = torch.randn(B, d, device=dev).to(torch.bfloat16); Cb = torch.randn(Nc, d, device=dev).to(torch.bfloat16)
    r = {"B": B, "Nc": Nc, "plan_us": [], "dc_alone_8p_us": [], "dc_us": t(lambda: kn.dc(G, Qb, 1.0)), "torch_dc_us": t(lambda: G.t() @ Qb)}
    outs = {}
    for rnd in range(2):
        for mode, key in ((0, "plan_us"), (1, "dc_alone_8p_us")):
            _lib.set_option("dc_alone_8p", mode)
            r[key].append(t(lambda: kn.inbatch_bwd(G, Qb, Cb, 1.0, go)))
            outs[mode] = kn.inbatch_bwd(G, Qb, Cb, 1.0, go)
    _lib.set_option("dc_alone_8p", 0)
    r["dC_max_rel_diff"] = ((outs[0][1] - outs[1][1]).abs().max() / outs[0][1].abs().max()).item()
    print(json.dumps(r), flush=True)
    del G, Qb, Cb, outs; torch.cuda.empty_cache()
