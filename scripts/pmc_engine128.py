#!/usr/bin/env python3
"""Workload for `scripts/gpu_run.sh pmc-engine` (VERDICT r5 #7 and #2c): the launches whose SQ counters round 7 needs side by side --
  * torch.matmul bf16 8192 x 8192 x 768 (the library GEMM the statistics GEMM is measured against),
  * dprhot_dq / dprhot_dc as separate launches at 8192 x 8192 x 768 and 4096 x 65536 x 768 (the 128 x 128 engine, where the hand-written
    GEMMs lose to torch by 1.3-1.5x: profiles/r05_bwd_long_axis.txt), and torch's own two GEMMs at the same shapes,
  * the statistics GEMM / dScores GEMM / backward pair of the library at 8192 x 8192 x 768 for reference.
Run under rocprofv3 --pmc (counters only, no trace); scripts/pmc_query.py prints per-kernel averages."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from dpr_scale_amd.hotpath import HipKernels  # noqa: E402

kn = HipKernels()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
REPS = int(os.environ.get("PMC_REPS", "4"))
A = torch.randn(8192, 768, device=dev).to(torch.bfloat16)
Bm = torch.randn(8192, 768, device=dev).to(torch.bfloat16)
for _ in range(REPS):
    torch.matmul(A, Bm.t())
for B, Nc, d in ((8192, 8192, 768), (4096, 65536, 768)):
    G = (torch.randn(B, Nc, device=dev) * 0.01).to(torch.bfloat16)
    Qb = torch.randn(B, d, device=dev).to(torch.bfloat16)
    Cb = torch.randn(Nc, d, device=dev).to(torch.bfloat16)
    go = torch.ones(1, device=dev)
    for _ in range(REPS):
        kn.dq(G, Cb, 1.0)
        kn.dc(G, Qb, 1.0)
        torch.matmul(G, Cb)
        torch.matmul(G.t(), Qb)
        if B == 8192:
            kn.inbatch_bwd(G, Qb, Cb, 1.0, go)
    del G, Qb, Cb
    torch.cuda.empty_cache()
y = torch.arange(8192, device=dev)
m8 = torch.zeros(8192, dtype=torch.uint8, device=dev)
for _ in range(REPS):
    kn.inbatch_fwd(A, Bm, y, 0, m8, 1.0, 1.0 / 8192, want_logits=False)
torch.cuda.synchronize()
