"""Round 6: long context axis -- the dQ units' context slices cut for ONE round of units on the 256 CUs (option dq_one_round = 1) against
at most 16 slices; us per dprhot_inbatch_bwd call, alternating; dQ compared (different slice boundaries: equal to fp32 rounding)."""
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
for B, Nc in ((1024, 32768), (1024, 65536), (2048, 65536), (1024, 49152), (1536, 65536), (1024, 65600)):
    G = (torch.randn(B, Nc, device=dev) * 0.01).to(torch.bfloat16); Qb = torch.randn(B, d, device=dev).to(torch.bfloat16); Cb = torch.randn(Nc, d, device=dev).to(torch.bfloat16)
    r = {"B": B, "Nc": Nc, "slices16_us": [], "one_round_us": []}
    outs = {}
    for rnd in range(2):
        for mode, key in ((0, "slices16_us"), (1, "one_round_us")):
            _lib.set_option("dq_one_round", mode)
            kn2 = HipKernels()  # (the workspace size depends on the plan)
            r[key].append(t(lambda: kn2.inbatch_bwd(G, Qb, Cb, 1.0, go)))
            outs[mode] = kn2.inbatch_bwd(G, Qb, Cb, 1.0, go)
    _lib.set_option("dq_one_round", 0)
    r["dQ_max_rel_diff"] = ((outs[0][0] - outs[1][0]).abs().max() / outs[0][0].abs().max()).item()
    ref = G.float() @ Cb.float()
    r["dQ_err_vs_fp32"] = [((outs[m][0] - ref).abs().max() / ref.abs().max()).item() for m in (0, 1)]
    print(json.dumps(r), flush=True)
    del G, Qb, Cb, outs, ref; torch.cuda.empty_cache()
