"""Round 6: dprhot_inbatch_bwd as the plan runs it (pair launch / long-axis rule) against its two GEMMs launched apart on the LDS-DMA
128 x 128 tile (options unfused_bwd = 1, g128_dma = 1), over the (B, Nc) grid; one process, alternating; us per call."""
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from dpr_scale_amd import _lib  # noqa: E402
from dpr_scale_amd.hotpath import HipKernels  # noqa: E402

kn = HipKernels()
dev = torch.device("cuda", 0)


def t(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)


d = 768
go = torch.ones(1, device=dev)
shapes = [(B, Nc) for B in (256, 512, 1024, 2048, 4096, 8192) for Nc in (4096, 8192, 16384, 32768, 65536) if B * Nc <= 8192 * 16384 * 2 and Nc >= B]
for B, Nc in shapes:
    G = (torch.randn(B, Nc, device=dev) * 0.01).to(torch.bfloat16)
    Qb = torch.randn(B, d, device=dev).to(torch.bfloat16)
    Cb = torch.randn(Nc, d, device=dev).to(torch.bfloat16)
    r = {"B": B, "Nc": Nc, "plan_us": [], "apart_dma_us": [], "plan_dma_us": []}
    for rnd in range(2):
        _lib.set_option("unfused_bwd", 0); _lib.set_option("g128_dma", 0)
        r["plan_us"].append(t(lambda: kn.inbatch_bwd(G, Qb, Cb, 1.0, go)))
        _lib.set_option("g128_dma", 1)
        r["plan_dma_us"].append(t(lambda: kn.inbatch_bwd(G, Qb, Cb, 1.0, go)))
        _lib.set_option("unfused_bwd", 1)
        r["apart_dma_us"].append(t(lambda: kn.inbatch_bwd(G, Qb, Cb, 1.0, go)))
    _lib.set_option("unfused_bwd", 0); _lib.set_option("g128_dma", 0)
    print(json.dumps(r), flush=True)
    del G, Qb, Cb
    torch.cuda.empty_cache()
