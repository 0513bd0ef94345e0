"""Interleaved A/B at 8192 x 8192 x 768, one process, alternating rounds (guide 5.4 rule 24): the library GEMM (torch.matmul), the
statistics GEMM, the dScores GEMM, the one-pass forward (option nl_p16 = 1) against the two-pass forward (nl_p16 = 0), the backward pair.
Every arm is timed as N back-to-back launches between two HIP events; rounds alternate so that all arms see the same thermal state."""
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from bench import HotPathStep, P  # noqa: E402
from dpr_scale_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
B = Nc = int(os.environ.get("AB_N", "8192"))
d = int(os.environ.get("AB_D", "768"))
N = int(os.environ.get("AB_LAUNCHES", "20"))
hp = HotPathStep(B, Nc // B, d, 1.0, 1, 0, dev)
hp.k_prep()
A = hp.Qb
Bt = hp.Cb.t()


def fwd(mode):
    def f():
        _lib.set_option("nl_p16", mode)
        hp.k_fwd()
    return f


arms = {"torch_matmul": lambda: torch.matmul(A, Bt), "stats_gemm": hp.k_sim, "dscores_gemm": hp.k_dscores, "fwd_one_pass": fwd(1),
        "fwd_two_pass": fwd(0), "backward_pair": hp.k_bwd}
for fn in arms.values():
    fn()
torch.cuda.synchronize()
res = {k: [] for k in arms}
for rnd in range(int(os.environ.get("AB_ROUNDS", "5"))):
    for name, fn in arms.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N):
            fn()
        e1.record()
        e1.synchronize()
        res[name].append(round(e0.elapsed_time(e1) * 1e3 / N, 1))
_lib.set_option("nl_p16", 1)
print(json.dumps({"shape": [B, Nc, d], "launches_per_sample": N, "us": res}))
