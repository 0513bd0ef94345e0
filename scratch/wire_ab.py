"""cfg3 per rank, the operator's call sequence (dprhot_train_step_packed_f32 + dprhot_rescale_grads) with fp32 or bf16 dC partials:
python scratch/wire_ab.py --kind 2|0 [under rocprofv3 --kernel-trace --stats: the per-kernel durations of ONE wire format]."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import HotPathStep, P, time_kernel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kind", type=int, default=2)
ap.add_argument("--eager", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)
B, K, d, W = 128, 8, 768, 8
hp = HotPathStep(B, K, d, 1.0, W, 0, dev, dist_mode=True)
hp.k_pack()
for r in range(W):
    hp.Cb[r * hp.rows_c:(r + 1) * hp.rows_c].copy_(hp.send)
lib, _lib = hp.lib, hp._lib
nsl = _lib.train_dq_slabs(B, hp.Nc, d)
part = torch.empty((max(nsl, 1), B, d), dtype=torch.float32, device=dev)
out2 = torch.empty(2, dtype=torch.float32, device=dev)
dCw = hp.dC if a.kind == 2 else torch.empty((hp.Nc, d), dtype=torch.bfloat16, device=dev)


def train_step():
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.dprhot_train_step_packed_f32(P(hp.q), P(hp.Cb), P(hp.Qb), B, W, 0, hp.n_ctx, d, P(hp.y), hp.inv_T, hp.gscale, 1.0 / hp.Nq,
                                          P(hp.go), P(hp.row_loss), P(hp.row_lse), P(hp.loss_sum), None, P(hp.dQ),
                                          P(part) if nsl > 0 else None, P(dCw), a.kind, P(hp.ws), hp.ws_bytes, st)
    rc = rc or lib.dprhot_rescale_grads(P(hp.dQ), hp.dQ.numel(), P(part) if nsl > 0 else None, nsl, P(dCw), dCw.numel(), a.kind,
                                        P(hp.go), P(hp.go), P(out2), st)
    assert rc == 0, _lib.lib.dprhot_last_error()


us = time_kernel(hp, train_step, reps=30, iters=10, use_graph=not a.eager)
print(json.dumps({"kind": a.kind, "step_us": round(us, 2)}), flush=True)
