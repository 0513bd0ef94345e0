#!/usr/bin/env python3
"""bench.py -- query-passage pairs/sec of dpr-scale's in-batch contrastive hot path on MI355X.

A "step" is one pass of the hot path over one synthetic NQ-shaped batch whose embeddings are already resident
in HBM: fp32->bf16 cast of q/c -> (W>1: RCCL all-gather of context rows + mask) -> sim (+mask, 1/T) ->
row-softmax CE + dScores -> loss sum (-> W>1: all-reduce) -> dC, dQ GEMMs (-> W>1: reduce-scatter of dC),
i.e. everything dpr_task.py:163-212 and its autograd backward do between the encoder outputs and their
gradients.  The step is driven through the C ABI (include/dprhot.h) exactly as a binding would drive it.

N=1 workload = BASELINE.json configs[1]: bert-base shapes, batch 32, 1 positive + 7 negatives, d=768, no
all-gather.  N>1: the same per-GPU batch on every rank (weak scaling; the global negatives grow with N).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--negatives 7] [--dim 768]
                  [--no-cpu-baseline] [--e2e]        (--e2e adds the bert-base end-to-end step as extra info)
Multi-GPU: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=32, help="queries per GPU")
    ap.add_argument("--negatives", type=int, default=7)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--e2e", action="store_true")
    return ap.parse_args()


def p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class HotPathStep:
    """Pre-allocated buffers + the C-ABI call sequence of one step on one rank."""

    def __init__(self, B, K, d, T, W, r, dev, group=None):
        from dpr_scale_amd import _lib
        from dpr_scale_amd import dist as D
        from oracle.inbatch_oracle import synth_embeddings  # input generator only (shared with the tests)

        self.lib, self._lib, self.D = _lib.lib, _lib, D
        self.B, self.K, self.d, self.T, self.W, self.r, self.group = B, K, d, T, W, r, group
        self.n_ctx = B * K
        self.Nc = W * self.n_ctx
        self.Nq = W * B
        q, c, y, m = synth_embeddings(1234 + r, B, K, d, "U", False)
        f32, bf16 = torch.float32, torch.bfloat16
        self.q = torch.from_numpy(q).to(dev)
        self.c = torch.from_numpy(c).to(dev)
        self.y = torch.from_numpy(y).to(dev)
        self.m8 = torch.from_numpy(m.astype(np.uint8)).to(dev)
        self.Qb = torch.empty((B, d), dtype=bf16, device=dev)
        self.Cb = torch.empty((self.Nc, d), dtype=bf16, device=dev)
        self.send = self.Cb[:self.n_ctx] if W == 1 else torch.empty((self.n_ctx, d), dtype=bf16, device=dev)
        self.mask_all = torch.zeros(self.Nc, dtype=torch.uint8, device=dev)
        self.row_loss = torch.empty(B, dtype=f32, device=dev)
        self.row_lse = torch.empty(B, dtype=f32, device=dev)
        self.loss_sum = torch.empty(1, dtype=f32, device=dev)
        self.G = torch.empty((B, self.Nc), dtype=bf16, device=dev)
        self.dQ = torch.empty((B, d), dtype=f32, device=dev)
        self.dC = torch.empty((self.Nc, d), dtype=f32, device=dev)
        self.dc = self.dC if W == 1 else torch.empty((self.n_ctx, d), dtype=f32, device=dev)
        self.go = torch.ones(1, dtype=f32, device=dev)
        self.ws_bytes = _lib.workspace_bytes(B, self.Nc, d)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        self.inv_T = 1.0 / T
        self.gscale = self.inv_T / self.Nq
        if W == 1:
            self.mask_all.copy_(self.m8)

    def stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    # individual stages (also timed one by one for the roofline report)
    def k_cast_q(self):
        self._lib.check(self.lib.dprhot_cast_bf16(p(self.q), p(self.Qb), self.q.numel(), self.stream()))

    def k_cast_c(self):
        self._lib.check(self.lib.dprhot_cast_bf16(p(self.c), p(self.send), self.c.numel(), self.stream()))

    def k_fwd(self):
        self._lib.check(self.lib.dprhot_inbatch_fwd(
            p(self.Qb), self.B, p(self.Cb), self.Nc, self.d, p(self.y), self.r * self.n_ctx, p(self.mask_all),
            self.inv_T, self.gscale, None, p(self.row_loss), p(self.row_lse), p(self.loss_sum), p(self.G),
            p(self.ws), self.ws_bytes, self.stream()))

    def k_bwd(self):
        self._lib.check(self.lib.dprhot_inbatch_bwd(
            p(self.G), p(self.Qb), p(self.Cb), self.B, self.Nc, self.d, 1.0, p(self.go), p(self.dQ), p(self.dC),
            p(self.ws), self.ws_bytes, self.stream()))

    def k_sim(self):
        self._lib.check(self.lib.dprhot_sim_fwd(p(self.Qb), self.B, p(self.Cb), self.Nc, self.d, p(self.mask_all),
                                                self.inv_T, p(self.ws), self.stream()))

    def k_softmax(self):
        self._lib.check(self.lib.dprhot_softmax_ce_fwd_bwd(
            p(self.ws), self.B, self.Nc, p(self.y), self.r * self.n_ctx, self.gscale, None, 0, p(self.row_loss),
            p(self.row_lse), p(self.G), self.stream()))

    def k_dq(self):
        self._lib.check(self.lib.dprhot_dq(p(self.G), p(self.Cb), self.B, self.Nc, self.d, 1.0, p(self.go), p(self.dQ),
                                           p(self.ws), self.ws_bytes, self.stream()))

    def k_dc(self):
        self._lib.check(self.lib.dprhot_dc(p(self.G), p(self.Qb), self.B, self.Nc, self.d, 1.0, p(self.go), p(self.dC),
                                           self.stream()))

    def step(self):
        self.k_cast_q()
        self.k_cast_c()
        if self.W > 1:
            self.D.all_gather_rows(self.send, self.Cb, self.group)
            self.D.all_gather_rows(self.m8, self.mask_all, self.group)
        self.k_fwd()
        if self.W > 1:
            self.D.all_reduce_sum(self.loss_sum, self.group)
        self.k_bwd()
        if self.W > 1:
            self.D.reduce_scatter_rows(self.dC, self.dc, self.group)


def time_kernel(fn, iters=200, warm=20):
    """Average duration of `fn`'s launches, HIP events on the launch stream."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters  # us


def cpu_baseline(B, K, d, T, budget_s=12.0):
    """The plain-C port of the same step (oracle/inbatch_oracle.c) on this box's host cores."""
    from oracle import c_oracle
    from oracle.inbatch_oracle import synth_embeddings

    lib = c_oracle.load()
    q, c, y, m = synth_embeddings(1234, B, K, d, "U", False)
    c_oracle.train_step(lib, q, c, y, 0, m, T, B)  # warm
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        c_oracle.train_step(lib, q, c, y, 0, m, T, B)
        n += 1
    el = time.perf_counter() - t0
    return {"value": B * n / el, "unit": "query-passage pairs/s", "cores": int(lib.oracle_num_threads()), "kind": "port",
            "sample": f"{n} steps of the same B={B} K={K} d={d} workload in {el:.1f} s (oracle/inbatch_oracle.c, OpenMP)"}


def main():
    a = parse()
    W = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if W > 1:
        assert a.gpus == W, f"--gpus {a.gpus} but WORLD_SIZE={W}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if W > 1:
        dist.init_process_group("nccl", device_id=dev)
    B, K, d, T = a.batch, 1 + a.negatives, a.dim, a.temperature
    hp = HotPathStep(B, K, d, T, W, rank, dev)

    for _ in range(a.warmup):
        hp.step()
    torch.cuda.synchronize()
    if W > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        hp.step()
    torch.cuda.synchronize()
    if W > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if W > 1:
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = tt.item()

    out = None
    if rank == 0:
        Nc, bn, bd, nd = hp.Nc, float(B) * hp.Nc, float(B) * d, float(hp.Nc) * d
        kern = {  # name: (fn, algorithmic bytes, flops)  -- DESIGN.md "algorithmic bytes per unit"
            "sim_fwd": (hp.k_sim, 2 * (bd + nd) + 4 * bn, 2 * bn * d),
            "softmax_ce_fwd_bwd": (hp.k_softmax, 6 * bn, 0.0),
            "dq": (hp.k_dq, 2 * bn + 2 * nd + 4 * bd, 2 * bn * d),
            "dc": (hp.k_dc, 2 * bn + 2 * bd + 4 * nd, 2 * bn * d),
            "cast_c": (hp.k_cast_c, 6 * float(hp.n_ctx) * d, 0.0),
        }
        ktimes = {}
        for name, (fn, by, fl) in kern.items():
            us = time_kernel(fn)
            ktimes[name] = {"us": round(us, 3), "GBps": round(by / us * 1e-3, 1), "TFLOPs": round(fl / us * 1e-6, 2)}
        dom = max(ktimes, key=lambda k: ktimes[k]["us"])
        by, fl = kern[dom][1], kern[dom][2]
        us = ktimes[dom]["us"]
        roof = {"kernel": dom, "bound": "hbm", "achieved": round(by / us * 1e-3, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(by / us * 1e-3 / HBM_PEAK_GBS, 5), "traffic": None,
                "avg_launch_us": us, "algorithmic_bytes": by}
        out = {
            "metric": "query-passage pairs/sec (in-batch contrastive hot path: gather+sim+softmax-CE+dQ/dC)",
            "value": round(W * B * a.steps / el, 1), "unit": "pairs/s", "n_gpus": W, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(el / a.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"cfg2-shaped per GPU: B={B} queries x (1+{a.negatives}) contexts, d={d}, T={T}; "
                                   f"global Nq={W * B}, Nc={hp.Nc}; embeddings resident in HBM, step driven through the C ABI",
                       "global_batch": W * B, "global_negatives_per_query": hp.Nc - 1, "parallelism": f"dp{W}"},
            "roofline": roof, "kernels": ktimes,
        }
        if not a.no_cpu_baseline and W == 1:
            out["cpu_baseline"] = cpu_baseline(B, K, d, T)
    if a.e2e and W == 1:
        try:
            from bench_e2e import end_to_end
            out["end_to_end"] = end_to_end(B, K, d, T, dev)
        except Exception as e:  # extra info only
            out["end_to_end"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(out))
    if W > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
