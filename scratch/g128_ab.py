"""A/B of option g128_dma (round 6: the 128 x 128 tile staged by LDS-DMA, csrc/gemm128d.h) on the launches that run on the 128 x 128
engine: dprhot_dq / dprhot_dc apart (long context axes), the stored-logits forward and the backward below 128 tiles.  One process,
arms alternating; us per call (HIP events around 10 calls)."""
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from bench import HotPathStep, time_kernel  # noqa: E402
from dpr_scale_amd import _lib  # noqa: E402
from dpr_scale_amd.hotpath import HipKernels  # noqa: E402

kn = HipKernels()
dev = torch.device("cuda", 0)


def t(fn, n=10):
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


for B, Nc, d in ((1024, 65536, 768), (8192, 8192, 768), (4096, 65536, 768), (512, 16384, 768), (2048, 32768, 768)):
    G = (torch.randn(B, Nc, device=dev) * 0.01).to(torch.bfloat16)
    Qb = torch.randn(B, d, device=dev).to(torch.bfloat16)
    Cb = torch.randn(Nc, d, device=dev).to(torch.bfloat16)
    r = {"B": B, "Nc": Nc, "d": d, "dq_us": {}, "dc_us": {}}
    for rnd in range(2):
        for mode in (0, 1):
            _lib.set_option("g128_dma", mode)
            r["dq_us"].setdefault(mode, []).append(t(lambda: kn.dq(G, Cb, 1.0)))
            r["dc_us"].setdefault(mode, []).append(t(lambda: kn.dc(G, Qb, 1.0)))
    r["torch_dq_us"] = t(lambda: G @ Cb)
    r["torch_dc_us"] = t(lambda: G.t() @ Qb)
    print(json.dumps(r), flush=True)
    del G, Qb, Cb
    torch.cuda.empty_cache()
for B, Nc in ((256, 8192), (512, 8192), (1024, 65536)):
    res = {"B": B, "Nc": Nc, "d": 768, "step_us": {}, "fwd_us": {}, "bwd_us": {}}
    hp = HotPathStep(B, Nc // B, 768, 1.0, 1, 0, dev)
    hp.k_prep()
    for rnd in range(2):
        for mode in (0, 1):
            _lib.set_option("g128_dma", mode)
            res["step_us"].setdefault(mode, []).append(round(time_kernel(hp, hp.k_step, reps=10, iters=3), 1))
            res["fwd_us"].setdefault(mode, []).append(round(time_kernel(hp, hp.k_fwd, reps=10, iters=3), 1))
            res["bwd_us"].setdefault(mode, []).append(round(time_kernel(hp, hp.k_bwd, reps=10, iters=3), 1))
    print(json.dumps(res), flush=True)
    del hp
    torch.cuda.empty_cache()
_lib.set_option("g128_dma", 0)
